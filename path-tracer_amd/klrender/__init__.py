"""Python face of the MI355X path-tracing backend: a thin ctypes layer over the C ABI
declared in include/Math/kl_render.h (library: path-tracer_amd/lib/libklrender.so).

There is no CPU fallback: if the HIP library is missing or a call fails, an exception is
raised. `Renderer` mirrors the state the reference host keeps around its render dispatch
(include/Rendering/Opengl/GLSL_Renderer.h:33-63): resolution, frame counter, view mode,
camera, plus the scene/texture uploads that replace the shader's constant arrays.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KLRENDER_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("KLRENDER_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libklrender.so")

MODE_AO, MODE_PT, MODE_Z = 0, 1, 2
DIFFUSE, GLASS, EMISSIVE, TEXTURED = 0, 1, 2, 3
TEX_HDRI, TEX_ALBEDO = 0, 1
TONEMAP_CLAMP, TONEMAP_ACES = 0, 1
ACCUM_MEAN, ACCUM_SUM, ACCUM_MEAN_UNORM8 = 0, 1, 2
ACCEL_AUTO, ACCEL_BRUTE, ACCEL_BVH = 0, 1, 2
BUILD_HOST_SAH, BUILD_GPU_LBVH = 0, 1
CONTINUE = 0xFFFFFFFF
MISS = 0xFFFFFFFF
NOALLOC_MAX_RAYS = 262144
SPLIT_TILES, SPLIT_SAMPLES, EXCHANGE_COPY = 0, 1, 0x100

material_dtype = np.dtype([("type", "<i4"), ("color", "<f4", 3), ("emissive", "<f4"),
                           ("roughness", "<f4"), ("ior", "<f4"), ("pad", "<f4")])
sphere_dtype = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", "<u4")])
quad_dtype = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3), ("v3", "<f4", 3), ("material", "<u4")])
tri_dtype = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3),
                      ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2), ("material", "<u4")])


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3),
                ("focal", C.c_float), ("sensor", C.c_float)]


class Params(C.Structure):
    _fields_ = [("max_bounces", C.c_uint32), ("max_frames", C.c_uint32), ("eps", C.c_float),
                ("max_dist", C.c_float), ("ao_length", C.c_float), ("hdri_strength", C.c_float),
                ("accel", C.c_uint32), ("frames_in_flight", C.c_uint32), ("fused_compaction", C.c_uint32),
                ("collect_stats", C.c_uint32), ("profile_kernels", C.c_uint32), ("batch_lanes", C.c_uint32),
                ("trace_primary_per_frame", C.c_uint32), ("bvh_builder", C.c_uint32), ("spp_grid", C.c_uint32), ("reserved", C.c_uint32 * 1)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("frames", "paths", "rays", "nodes", "sphere_tests", "quad_tests",
                                          "tri_tests", "lane_steps_active", "lane_steps_total")]


class Timing(C.Structure):
    _fields_ = [("intersect_ms", C.c_double), ("intersect_launches", C.c_uint64),
                ("render_ms", C.c_double), ("render_calls", C.c_uint64),
                ("shade_ms", C.c_double), ("shade_launches", C.c_uint64),
                ("generate_ms", C.c_double), ("generate_launches", C.c_uint64),
                ("accumulate_ms", C.c_double), ("accumulate_launches", C.c_uint64)]


class BvhStats(C.Structure):
    _fields_ = [("nodes", C.c_uint32), ("leaves", C.c_uint32), ("max_depth", C.c_uint32), ("max_leaf_size", C.c_uint32),
                ("spheres", C.c_uint32), ("quads", C.c_uint32), ("tris", C.c_uint32), ("sah_cost", C.c_float),
                ("build_seconds", C.c_double), ("node_bytes", C.c_uint64), ("prim_bytes", C.c_uint64),
                ("flat_spheres", C.c_uint32), ("flat_quads", C.c_uint32), ("flat_tris", C.c_uint32), ("builder", C.c_uint32),
                ("refit_seconds", C.c_double)]


class SynthParams(C.Structure):
    _fields_ = [("blobs", C.c_uint32), ("tris_per_blob", C.c_uint32), ("analytic_spheres", C.c_uint32), ("seed", C.c_uint32)]


# every entry point include/Math/kl_render.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "kl_create": (C.c_int32, [C.POINTER(_P), C.c_int32, C.c_uint32, C.c_uint32]),
    "kl_destroy": (C.c_int32, [_P]),
    "kl_resize": (C.c_int32, [_P, C.c_uint32, C.c_uint32]),
    "kl_recompile": (C.c_int32, [_P]),
    "kl_set_camera": (C.c_int32, [_P, C.POINTER(Camera)]),
    "kl_set_mode": (C.c_int32, [_P, C.c_uint32]),
    "kl_default_params": (C.c_int32, [C.POINTER(Params)]),
    "kl_default_camera": (C.c_int32, [C.POINTER(Camera)]),
    "kl_set_params": (C.c_int32, [_P, C.POINTER(Params)]),
    "kl_get_params": (C.c_int32, [_P, C.POINTER(Params)]),
    "kl_reset": (C.c_int32, [_P]),
    "kl_get_frame": (C.c_int32, [_P, C.POINTER(C.c_uint32)]),
    "kl_upload_texture": (C.c_int32, [_P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "kl_upload_scene": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32]),
    "kl_load_builtin_scene": (C.c_int32, [_P]),
    "kl_refit_scene": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32]),
    "kl_get_bvh_stats": (C.c_int32, [_P, C.POINTER(BvhStats)]),
    "kl_synth_scene_counts": (C.c_int32, [C.POINTER(SynthParams)] + [C.POINTER(C.c_uint32)] * 4),
    "kl_synth_scene_fill": (C.c_int32, [C.POINTER(SynthParams), _P, _P, _P, _P]),
    "kl_render_frames": (C.c_int32, [_P, C.c_uint32, C.c_uint32]),
    "kl_synchronize": (C.c_int32, [_P]),
    "kl_read_raw": (C.c_int32, [_P, _P, C.c_size_t]),
    "kl_read_accum": (C.c_int32, [_P, _P, C.c_size_t]),
    "kl_resolve": (C.c_int32, [_P, C.c_uint32, _P, C.c_size_t]),
    "kl_get_counters": (C.c_int32, [_P, C.POINTER(Counters)]),
    "kl_reset_counters": (C.c_int32, [_P]),
    "kl_get_timing": (C.c_int32, [_P, C.POINTER(Timing)]),
    "kl_trace_rays": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, _P, _P]),
    "kl_get_stream": (C.c_int32, [_P, _P]),
    "kl_scene_blob_bytes": (C.c_int32, [_P, C.POINTER(C.c_size_t)]),
    "kl_scene_export": (C.c_int32, [_P, _P, C.c_size_t]),
    "kl_scene_import": (C.c_int32, [_P, _P, C.c_size_t]),
    "kl_trace_rays_noalloc": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, _P, _P]),
    "kl_set_shard": (C.c_int32, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "kl_get_local_rows": (C.c_int32, [_P, C.POINTER(C.c_uint32)]),
    "kl_set_accum_mode": (C.c_int32, [_P, C.c_uint32]),
    "kl_copy_accum_to_device": (C.c_int32, [_P, _P, C.c_size_t]),
    "kl_copy_accum_from_device": (C.c_int32, [_P, _P, C.c_size_t]),
    "kl_upload_env_float": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "kl_load_mesh": (C.c_int32, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint32)]),
    "kl_load_mesh_error": (C.c_char_p, []),
    "kl_free": (None, [_P]),
    "kl_checkpoint_save": (C.c_int32, [_P, C.c_char_p]),
    "kl_checkpoint_load": (C.c_int32, [_P, C.c_char_p]),
    "kl_group_create": (C.c_int32, [C.POINTER(_P), C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "kl_group_destroy": (C.c_int32, [_P]),
    "kl_group_size": (C.c_int32, [_P, C.POINTER(C.c_uint32)]),
    "kl_group_member": (C.c_int32, [_P, C.c_uint32, C.POINTER(_P)]),
    "kl_group_upload_scene": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32]),
    "kl_group_load_builtin_scene": (C.c_int32, [_P]),
    "kl_group_upload_texture": (C.c_int32, [_P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "kl_group_set_params": (C.c_int32, [_P, C.POINTER(Params)]),
    "kl_group_set_camera": (C.c_int32, [_P, C.POINTER(Camera)]),
    "kl_group_set_mode": (C.c_int32, [_P, C.c_uint32]),
    "kl_group_reset": (C.c_int32, [_P]),
    "kl_group_render_frames": (C.c_int32, [_P, C.c_uint32, C.c_uint32]),
    "kl_group_gather": (C.c_int32, [_P, _P, C.c_size_t]),
    "kl_group_image_device": (C.c_int32, [_P, C.c_uint32, C.POINTER(_P)]),
    "kl_group_last_error": (C.c_char_p, [_P]),
    "kl_status_string": (C.c_char_p, [C.c_int32]),
    "kl_last_error": (C.c_char_p, [_P]),
    "kl_abi_version": (C.c_uint32, []),
}

_lib = None


class KlError(RuntimeError):
    pass


def lib():
    """Loads libklrender.so; raises if it is absent (there is no other implementation). A process that also uses PyTorch on
    the GPU imports torch BEFORE the first call here: torch ships its own copy of the HIP runtime and cannot initialise once
    the system's copy, which this library names, is in the process (the other order works for both)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KlError("HIP library missing: %s (build it with `make` or __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def default_params(**kw):
    p = Params()
    lib().kl_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_camera():
    c = Camera()
    lib().kl_default_camera(C.byref(c))
    return c


def synth_scene(blobs, tris_per_blob=1000, analytic_spheres=0, seed=1):
    """Arrays of the synthetic benchmark scene: (spheres, quads, tris, materials)."""
    sp = SynthParams(blobs, tris_per_blob, analytic_spheres, seed)
    n = [C.c_uint32() for _ in range(4)]
    rc = lib().kl_synth_scene_counts(C.byref(sp), *[C.byref(x) for x in n])
    if rc != 0:
        raise KlError("kl_synth_scene_counts: %s" % lib().kl_status_string(rc).decode())
    s = np.zeros(n[0].value, dtype=sphere_dtype)
    q = np.zeros(n[1].value, dtype=quad_dtype)
    t = np.zeros(n[2].value, dtype=tri_dtype)
    m = np.zeros(n[3].value, dtype=material_dtype)
    rc = lib().kl_synth_scene_fill(C.byref(sp), _ptr(s), _ptr(q), _ptr(t), _ptr(m))
    if rc != 0:
        raise KlError("kl_synth_scene_fill: %s" % lib().kl_status_string(rc).decode())
    return s, q, t, m


def load_mesh(path, material=0, n_materials=0):
    """Triangles of a Wavefront .obj or Stanford .ply file as a tri_dtype array (kl_load_mesh)."""
    buf = _P()
    n = C.c_uint32()
    rc = lib().kl_load_mesh(os.fsencode(path), material, n_materials, C.byref(buf), C.byref(n))
    if rc != 0:
        raise KlError("kl_load_mesh: %s (%s)" % (lib().kl_status_string(rc).decode(), lib().kl_load_mesh_error().decode()))
    try:
        out = np.zeros(n.value, dtype=tri_dtype)
        if n.value:
            C.memmove(out.ctypes.data, buf, out.nbytes)
    finally:
        lib().kl_free(buf)
    return out


class Renderer:
    """One context = one GPU + one stream. Not thread-safe, like the reference's GL context."""

    def __init__(self, width, height, device=0):
        self._h = _P()
        rc = lib().kl_create(C.byref(self._h), device, width, height)
        if rc != 0:
            raise KlError("kl_create failed: %s" % lib().kl_status_string(rc).decode())
        self.width, self.height, self.device = width, height, device
        self.shard = (0, 1, 1)
        self.accum_mode = ACCUM_MEAN

    # -- plumbing
    def _ck(self, rc, what):
        if rc != 0:
            msg = lib().kl_last_error(self._h)
            raise KlError("%s: %s (%s)" % (what, lib().kl_status_string(rc).decode(), msg.decode() if msg else ""))

    def close(self):
        if getattr(self, "_h", None):
            lib().kl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- uniforms and state
    def resize(self, width, height):
        self._ck(lib().kl_resize(self._h, width, height), "kl_resize")
        self.width, self.height = width, height

    def recompile(self):
        self._ck(lib().kl_recompile(self._h), "kl_recompile")

    def set_camera(self, cam):
        self._ck(lib().kl_set_camera(self._h, C.byref(cam)), "kl_set_camera")

    def set_mode(self, mode):
        self._ck(lib().kl_set_mode(self._h, mode), "kl_set_mode")

    def set_params(self, params):
        self._ck(lib().kl_set_params(self._h, C.byref(params)), "kl_set_params")

    def get_params(self):
        p = Params()
        self._ck(lib().kl_get_params(self._h, C.byref(p)), "kl_get_params")
        return p

    def update_params(self, **kw):
        p = self.get_params()
        for k, v in kw.items():
            setattr(p, k, v)
        self.set_params(p)

    def reset(self):
        self._ck(lib().kl_reset(self._h), "kl_reset")

    @property
    def frame(self):
        f = C.c_uint32()
        self._ck(lib().kl_get_frame(self._h, C.byref(f)), "kl_get_frame")
        return f.value

    # -- uploads
    def upload_texture(self, which, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        if px.ndim == 2:
            px = px[:, :, None]
        h, w, ch = px.shape
        self._ck(lib().kl_upload_texture(self._h, which, _ptr(px), w, h, ch), "kl_upload_texture")

    def upload_scene(self, spheres=None, quads=None, tris=None, materials=None):
        def arr(a, dt):
            return np.ascontiguousarray(a, dtype=dt) if a is not None else np.zeros(0, dtype=dt)
        s, q, t, m = arr(spheres, sphere_dtype), arr(quads, quad_dtype), arr(tris, tri_dtype), arr(materials, material_dtype)
        self._ck(lib().kl_upload_scene(self._h, _ptr(s), len(s), _ptr(q), len(q), _ptr(t), len(t), _ptr(m), len(m)), "kl_upload_scene")

    def refit_scene(self, spheres=None, quads=None, tris=None):
        """The uploaded scene's primitives, moved: same counts and order; the tree keeps its shape (kl_refit_scene)."""
        def arr(a, dt):
            return np.ascontiguousarray(a, dtype=dt) if a is not None else np.zeros(0, dtype=dt)
        s, q, t = arr(spheres, sphere_dtype), arr(quads, quad_dtype), arr(tris, tri_dtype)
        self._ck(lib().kl_refit_scene(self._h, _ptr(s), len(s), _ptr(q), len(q), _ptr(t), len(t)), "kl_refit_scene")

    def load_builtin_scene(self):
        self._ck(lib().kl_load_builtin_scene(self._h), "kl_load_builtin_scene")

    def bvh_stats(self):
        st = BvhStats()
        self._ck(lib().kl_get_bvh_stats(self._h, C.byref(st)), "kl_get_bvh_stats")
        return st

    # -- the frame loop
    def render_frames(self, n, first_frame=CONTINUE):
        self._ck(lib().kl_render_frames(self._h, n, first_frame), "kl_render_frames")

    def synchronize(self):
        self._ck(lib().kl_synchronize(self._h), "kl_synchronize")

    # -- results
    @property
    def local_rows(self):
        r = C.c_uint32()
        self._ck(lib().kl_get_local_rows(self._h, C.byref(r)), "kl_get_local_rows")
        return r.value

    def read_raw(self):
        out = np.empty((self.local_rows, self.width, 4), dtype=np.float32)
        self._ck(lib().kl_read_raw(self._h, _ptr(out), out.nbytes), "kl_read_raw")
        return out

    def read_accum(self):
        out = np.empty((self.local_rows, self.width, 4), dtype=np.float32)
        self._ck(lib().kl_read_accum(self._h, _ptr(out), out.nbytes), "kl_read_accum")
        return out

    def resolve(self, tonemap=TONEMAP_CLAMP):
        out = np.empty((self.local_rows, self.width, 4), dtype=np.uint8)
        self._ck(lib().kl_resolve(self._h, tonemap, _ptr(out), out.nbytes), "kl_resolve")
        return out

    def counters(self):
        c = Counters()
        self._ck(lib().kl_get_counters(self._h, C.byref(c)), "kl_get_counters")
        return c

    def reset_counters(self):
        self._ck(lib().kl_reset_counters(self._h), "kl_reset_counters")

    def upload_env_float(self, pixels):
        """float rgb(a) equirectangular environment in place of the 8-bit one (None: back to 8-bit)"""
        if pixels is None:
            self._ck(lib().kl_upload_env_float(self._h, None, 0, 0, 0), "kl_upload_env_float")
            return
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        h, w, ch = px.shape
        self._ck(lib().kl_upload_env_float(self._h, _ptr(px), w, h, ch), "kl_upload_env_float")

    def save_checkpoint(self, path):
        self._ck(lib().kl_checkpoint_save(self._h, os.fsencode(path)), "kl_checkpoint_save")

    def load_checkpoint(self, path):
        self._ck(lib().kl_checkpoint_load(self._h, os.fsencode(path)), "kl_checkpoint_load")

    def timing(self):
        t = Timing()
        self._ck(lib().kl_get_timing(self._h, C.byref(t)), "kl_get_timing")
        return t

    def trace_rays(self, o, d):
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        n = len(o)
        t = np.empty(n, dtype=np.float32)
        prim = np.empty(n, dtype=np.uint32)
        uv = np.empty((n, 2), dtype=np.float32)
        self._ck(lib().kl_trace_rays(self._h, _ptr(o), _ptr(d), n, _ptr(t), _ptr(prim), _ptr(uv)), "kl_trace_rays")
        return t, prim, uv

    def scene_blob_bytes(self):
        n = C.c_size_t()
        self._ck(lib().kl_scene_blob_bytes(self._h, C.byref(n)), "kl_scene_blob_bytes")
        return n.value

    def scene_export(self, device_ptr, nbytes):
        """the packed scene (records, tree, shading records, flat-quad table) into device memory at device_ptr"""
        self._ck(lib().kl_scene_export(self._h, C.c_void_p(device_ptr), nbytes), "kl_scene_export")

    def scene_import(self, device_ptr, nbytes):
        """a block written by scene_export (on any device) becomes this context's scene"""
        self._ck(lib().kl_scene_import(self._h, C.c_void_p(device_ptr), nbytes), "kl_scene_import")

    def stream(self):
        """the context's HIP stream as an integer handle (kl_get_stream)"""
        h = C.c_void_p()
        self._ck(lib().kl_get_stream(self._h, C.byref(h)), "kl_get_stream")
        return h.value or 0

    def trace_rays_noalloc(self, o_dev, d_dev, n, t_dev, prim_dev, uv_dev):
        """kl_trace_rays_noalloc: DEVICE pointers (integers) in and out, at most NOALLOC_MAX_RAYS rays, asynchronous"""
        self._ck(lib().kl_trace_rays_noalloc(self._h, C.c_void_p(o_dev), C.c_void_p(d_dev), n, C.c_void_p(t_dev),
                                             C.c_void_p(prim_dev), C.c_void_p(uv_dev)), "kl_trace_rays_noalloc")

    # -- partitioning
    def set_shard(self, rank, world, band_rows):
        self._ck(lib().kl_set_shard(self._h, rank, world, band_rows), "kl_set_shard")
        self.shard = (rank, world, band_rows)

    def set_accum_mode(self, mode):
        self._ck(lib().kl_set_accum_mode(self._h, mode), "kl_set_accum_mode")
        self.accum_mode = mode

    def copy_accum_to_device(self, device_ptr, nbytes):
        self._ck(lib().kl_copy_accum_to_device(self._h, C.c_void_p(device_ptr), nbytes), "kl_copy_accum_to_device")

    def copy_accum_from_device(self, device_ptr, nbytes):
        self._ck(lib().kl_copy_accum_from_device(self._h, C.c_void_p(device_ptr), nbytes), "kl_copy_accum_from_device")


class Group:
    """Several GPUs from one process (kl_group_*): one context per device, image tiles or sample ranges, the exchange over
    RCCL (or, with EXCHANGE_COPY, by device copies: members may then share a GPU)."""

    def __init__(self, devices, split, width, height, band_rows=8):
        self._h = _P()
        devs = (C.c_int32 * len(devices))(*devices)
        rc = lib().kl_group_create(C.byref(self._h), devs, len(devices), split, width, height, band_rows)
        if rc != 0:
            raise KlError("kl_group_create failed: %s (%s)" % (lib().kl_status_string(rc).decode(), (lib().kl_group_last_error(None) or b"").decode()))
        self.width, self.height, self.n = width, height, len(devices)

    def _ck(self, rc, what):
        if rc != 0:
            raise KlError("%s: %s (%s)" % (what, lib().kl_status_string(rc).decode(), (lib().kl_group_last_error(self._h) or b"").decode()))

    def close(self):
        if getattr(self, "_h", None):
            lib().kl_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, spheres=None, quads=None, tris=None, materials=None):
        def arr(a, dt):
            return np.ascontiguousarray(a, dtype=dt) if a is not None else np.zeros(0, dtype=dt)
        s, q, t, m = arr(spheres, sphere_dtype), arr(quads, quad_dtype), arr(tris, tri_dtype), arr(materials, material_dtype)
        self._ck(lib().kl_group_upload_scene(self._h, _ptr(s), len(s), _ptr(q), len(q), _ptr(t), len(t), _ptr(m), len(m)), "kl_group_upload_scene")

    def load_builtin_scene(self):
        self._ck(lib().kl_group_load_builtin_scene(self._h), "kl_group_load_builtin_scene")

    def upload_texture(self, which, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        if px.ndim == 2:
            px = px[:, :, None]
        h, w, ch = px.shape
        self._ck(lib().kl_group_upload_texture(self._h, which, _ptr(px), w, h, ch), "kl_group_upload_texture")

    def set_params(self, params):
        self._ck(lib().kl_group_set_params(self._h, C.byref(params)), "kl_group_set_params")

    def set_camera(self, cam):
        self._ck(lib().kl_group_set_camera(self._h, C.byref(cam)), "kl_group_set_camera")

    def set_mode(self, mode):
        self._ck(lib().kl_group_set_mode(self._h, mode), "kl_group_set_mode")

    def reset(self):
        self._ck(lib().kl_group_reset(self._h), "kl_group_reset")

    def render_frames(self, n, first_frame=CONTINUE):
        self._ck(lib().kl_group_render_frames(self._h, n, first_frame), "kl_group_render_frames")

    def gather(self, to_host=True):
        out = np.empty((self.height, self.width, 4), dtype=np.float32) if to_host else None
        self._ck(lib().kl_group_gather(self._h, _ptr(out) if to_host else None, out.nbytes if to_host else 0), "kl_group_gather")
        return out
