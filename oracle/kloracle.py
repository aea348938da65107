"""ctypes binding of the CPU oracle (oracle/libkloracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The product package never imports this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MODE_AO, MODE_PT, MODE_Z = 0, 1, 2
DIFFUSE, GLASS, EMISSIVE, TEXTURED = 0, 1, 2, 3
TEX_HDRI, TEX_ALBEDO = 0, 1
TONEMAP_CLAMP, TONEMAP_ACES = 0, 1

# numpy views of the C structs (identical layout to the product's upload structs)
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
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mode", C.c_uint32),
                ("max_bounces", C.c_uint32), ("max_frames", C.c_uint32),
                ("eps", C.c_float), ("max_dist", C.c_float), ("ao_length", C.c_float),
                ("hdri_strength", C.c_float), ("use_bvh", C.c_int32), ("spp", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "rays", "nodes", "sphere_tests", "quad_tests", "tri_tests")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libkloracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libkloracle.so is missing: run `make -C oracle`")
        L = C.CDLL(path)
        L.ko_scene_create.restype = C.c_void_p
        L.ko_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.ko_scene_builtin.restype = C.c_void_p
        L.ko_scene_destroy.argtypes = [C.c_void_p]
        L.ko_scene_set_texture.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ko_scene_set_env_float.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ko_render_raw.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int]
        L.ko_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int]
        L.ko_accumulate_unorm8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int]
        L.ko_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.ko_rand4.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float * 4)]
        L.ko_hash.argtypes = [C.POINTER(C.c_uint32 * 4), C.POINTER(C.c_uint32 * 4)]
        L.ko_camera_ray.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                    C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3)]
        for f in ("ko_sinf", "ko_cosf", "ko_acosf"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_float]
        L.ko_atan2f.restype = C.c_float
        L.ko_atan2f.argtypes = [C.c_float, C.c_float]
        L.ko_env_uv.argtypes = [C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 2)]
        L.ko_trace.restype = C.c_int64
        L.ko_trace.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3),
                               C.POINTER(C.c_float), C.POINTER(C.c_float * 2)]
        L.ko_sphere_intersect.argtypes = [C.POINTER(C.c_float * 3)] * 3 + [C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.ko_quad_intersect.argtypes = [C.POINTER(C.c_float * 3)] * 6 + [C.POINTER(C.c_float), C.POINTER(C.c_float * 2)]
        L.ko_tri_intersect.argtypes = [C.POINTER(C.c_float * 3)] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_float * 2)]
        L.ko_default_params.argtypes = [C.POINTER(Params)]
        L.ko_default_camera.argtypes = [C.POINTER(Camera)]
        _lib = L
    return _lib


def default_params(**kw):
    p = Params()
    lib().ko_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_camera():
    c = Camera()
    lib().ko_default_camera(C.byref(c))
    return c


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Scene:
    def __init__(self, spheres=None, quads=None, tris=None, materials=None, builtin=False):
        L = lib()
        if builtin:
            self.h = L.ko_scene_builtin()
        else:
            def arr(a, dt):
                return np.ascontiguousarray(a, dtype=dt) if a is not None else np.zeros(0, dtype=dt)
            s, q, t, m = arr(spheres, sphere_dtype), arr(quads, quad_dtype), arr(tris, tri_dtype), arr(materials, material_dtype)
            self.h = L.ko_scene_create(_ptr(s), len(s), _ptr(q), len(q), _ptr(t), len(t), _ptr(m), len(m))
        if not self.h:
            raise RuntimeError("ko_scene_create failed")

    def set_texture(self, which, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        if px.ndim == 2:
            px = px[:, :, None]
        h, w, ch = px.shape
        if lib().ko_scene_set_texture(self.h, which, _ptr(px), w, h, ch) != 0:
            raise ValueError("bad texture")

    def set_env_float(self, pixels):
        if pixels is None:
            lib().ko_scene_set_env_float(self.h, None, 0, 0, 0)
            return
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        h, w, ch = px.shape
        if lib().ko_scene_set_env_float(self.h, _ptr(px), w, h, ch) != 0:
            raise ValueError("bad float environment")

    def render_raw(self, cam, params, frame, last_acc=None, rows=None, nthreads=None, counters=None):
        w, h = params.width, params.height
        out = np.zeros((h, w, 4), dtype=np.float32)
        y0, y1 = rows if rows else (0, h)
        nthreads = nthreads or min(8, os.cpu_count() or 1)
        la = np.ascontiguousarray(last_acc, dtype=np.float32) if last_acc is not None else None
        rc = lib().ko_render_raw(self.h, C.byref(cam), C.byref(params), frame, y0, y1, _ptr(out), _ptr(la),
                                 C.byref(counters) if counters is not None else None, nthreads)
        if rc != 0:
            raise RuntimeError("ko_render_raw failed: %d" % rc)
        return out

    def trace(self, params, o, d):
        t = C.c_float()
        uv = (C.c_float * 2)()
        prim = lib().ko_trace(self.h, C.byref(params), (C.c_float * 3)(*o), (C.c_float * 3)(*d), C.byref(t), C.byref(uv))
        return prim, t.value, (uv[0], uv[1])

    def close(self):
        if self.h:
            lib().ko_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def accumulate(acc, raw, frame, camera_change=False, unorm8=False):
    assert acc.dtype == np.float32 and raw.dtype == np.float32 and acc.flags.c_contiguous and raw.flags.c_contiguous
    fn = lib().ko_accumulate_unorm8 if unorm8 else lib().ko_accumulate
    fn(_ptr(acc), _ptr(raw), acc.size // 4, frame, int(camera_change))
    return acc


def render_mean(scene, cam, params, nframes, first_frame=0, nthreads=None, counters=None, unorm8=False):
    """Reference frame loop: raw -> running mean, frames first_frame..first_frame+nframes-1. unorm8: through RGBA8
    attachments, as the application ships (raw frame and mean rounded to 8 bits every frame)."""
    acc = np.zeros((params.height, params.width, 4), dtype=np.float32)
    for f in range(first_frame, first_frame + nframes):
        raw = scene.render_raw(cam, params, f, last_acc=acc, nthreads=nthreads, counters=counters)
        accumulate(acc, raw, f, camera_change=(f == 0), unorm8=unorm8)
    return acc


def resolve(acc, tonemap=TONEMAP_CLAMP):
    a = np.ascontiguousarray(acc, dtype=np.float32)
    out = np.zeros(a.shape[:-1] + (4,), dtype=np.uint8)
    lib().ko_resolve(_ptr(a), _ptr(out), a.size // 4, tonemap)
    return out


def rand4(x, y, frame):
    out = (C.c_float * 4)()
    lib().ko_rand4(x, y, frame, C.byref(out))
    return np.array(out[:], dtype=np.float32)


def hash4(seed):
    s = (C.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in seed])
    out = (C.c_uint32 * 4)()
    lib().ko_hash(C.byref(s), C.byref(out))
    return np.array(out[:], dtype=np.uint32)


def camera_ray(cam, w, h, x, y, subpixel):
    o = (C.c_float * 3)()
    d = (C.c_float * 3)()
    lib().ko_camera_ray(C.byref(cam), w, h, x, y, int(subpixel), C.byref(o), C.byref(d))
    return np.array(o[:], dtype=np.float32), np.array(d[:], dtype=np.float32)
