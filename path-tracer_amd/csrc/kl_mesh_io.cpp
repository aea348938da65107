// Scene ingestion: Wavefront OBJ and Stanford PLY files -> kl_tri[] for kl_upload_scene.
// The reference has mesh containers on the CPU side only (include/Object/Mesh.h:5-26: Tri, Quad and the
// empty Ngon / Mesh) and no loader; this follows their shape: polygons become triangle fans, texture
// coordinates ride along per corner, everything else in the files (normals, groups, material libraries,
// colours) is skipped.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "Math/kl_render.h"

namespace {

struct V3 { float x, y, z; };
struct V2 { float u, v; };
constexpr size_t kMaxElements = (size_t)1 << 27;    // vertices or faces of one file
constexpr size_t kMaxTriangles = (size_t)1 << 25;   // the scene limit of kl_upload_scene

void put(kl_tri &t, const V3 &a, const V3 &b, const V3 &c, const V2 &ta, const V2 &tb, const V2 &tc, uint32_t material) {
    t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z;
    t.v1[0] = b.x; t.v1[1] = b.y; t.v1[2] = b.z;
    t.v2[0] = c.x; t.v2[1] = c.y; t.v2[2] = c.z;
    t.uv0[0] = ta.u; t.uv0[1] = ta.v; t.uv1[0] = tb.u; t.uv1[1] = tb.v; t.uv2[0] = tc.u; t.uv2[1] = tc.v;
    t.material = material;
}

// "f" corner: v, v/vt, v//vn or v/vt/vn; 1-based, negative = relative to the end
bool obj_corner(const std::string &tok, long nv, long nt, long &vi, long &ti) {
    vi = ti = 0;
    const char *p = tok.c_str(); char *e = nullptr;
    long a = std::strtol(p, &e, 10);
    if (e == p) return false;
    vi = a < 0 ? nv + a : a - 1;
    ti = -1;
    if (*e == '/') {
        p = e + 1;
        if (*p != '/' && *p != '\0') {
            long b = std::strtol(p, &e, 10);
            if (e != p) ti = b < 0 ? nt + b : b - 1;
        }
    }
    return vi >= 0 && vi < nv && (ti == -1 || (ti >= 0 && ti < nt));
}

bool load_obj(const char *path, uint32_t material, uint32_t n_materials, std::vector<kl_tri> &out, std::string &err) {
    std::ifstream in(path);
    if (!in) { err = std::string("cannot read ") + path; return false; }
    std::vector<V3> v; std::vector<V2> vt;
    std::vector<std::string> mtl_names;          // `usemtl` names in order of first use -> material, material + 1, ...
    uint32_t cur = material;
    std::string line, kw;
    size_t lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        std::istringstream ls(line);
        if (!(ls >> kw) || kw[0] == '#') continue;
        if (kw == "v") { V3 p{0, 0, 0}; ls >> p.x >> p.y >> p.z; v.push_back(p); }
        else if (kw == "vt") { V2 t{0, 0}; ls >> t.u >> t.v; vt.push_back(t); }
        else if (kw == "usemtl") {
            std::string name; ls >> name;
            size_t k = 0;
            while (k < mtl_names.size() && mtl_names[k] != name) k++;
            if (k == mtl_names.size()) mtl_names.push_back(name);
            cur = material + (uint32_t)k;
            if (n_materials && cur >= n_materials) cur = n_materials - 1;      // more names than materials: the last one
        } else if (kw == "f") {
            std::vector<long> vi, ti; std::string tok;
            while (ls >> tok) {
                long a, b;
                if (!obj_corner(tok, (long)v.size(), (long)vt.size(), a, b)) {
                    err = std::string(path) + ":" + std::to_string(lineno) + ": bad face corner '" + tok + "'"; return false;
                }
                vi.push_back(a); ti.push_back(b);
            }
            if (vi.size() < 3) { err = std::string(path) + ":" + std::to_string(lineno) + ": face with fewer than 3 corners"; return false; }
            const V2 zero{0, 0};
            if (out.size() + vi.size() > kMaxTriangles) { err = "more than 2^25 triangles"; return false; }
            for (size_t k = 1; k + 1 < vi.size(); k++) {       // triangle fan
                kl_tri t;
                put(t, v[vi[0]], v[vi[k]], v[vi[k + 1]], ti[0] >= 0 ? vt[ti[0]] : zero, ti[k] >= 0 ? vt[ti[k]] : zero,
                    ti[k + 1] >= 0 ? vt[ti[k + 1]] : zero, cur);
                out.push_back(t);
            }
        }
    }
    return true;
}

enum PlyType { P_I8, P_U8, P_I16, P_U16, P_I32, P_U32, P_F32, P_F64, P_BAD };
PlyType ply_type(const std::string &s) {
    if (s == "char" || s == "int8") return P_I8;
    if (s == "uchar" || s == "uint8") return P_U8;
    if (s == "short" || s == "int16") return P_I16;
    if (s == "ushort" || s == "uint16") return P_U16;
    if (s == "int" || s == "int32") return P_I32;
    if (s == "uint" || s == "uint32") return P_U32;
    if (s == "float" || s == "float32") return P_F32;
    if (s == "double" || s == "float64") return P_F64;
    return P_BAD;
}
size_t ply_size(PlyType t) { static const size_t n[] = {1, 1, 2, 2, 4, 4, 4, 8, 0}; return n[t]; }
double ply_value(const unsigned char *p, PlyType t) {
    switch (t) {
        case P_I8: { int8_t x; std::memcpy(&x, p, 1); return x; }
        case P_U8: { uint8_t x; std::memcpy(&x, p, 1); return x; }
        case P_I16: { int16_t x; std::memcpy(&x, p, 2); return x; }
        case P_U16: { uint16_t x; std::memcpy(&x, p, 2); return x; }
        case P_I32: { int32_t x; std::memcpy(&x, p, 4); return x; }
        case P_U32: { uint32_t x; std::memcpy(&x, p, 4); return x; }
        case P_F32: { float x; std::memcpy(&x, p, 4); return x; }
        case P_F64: { double x; std::memcpy(&x, p, 8); return x; }
        default: return 0.0;
    }
}

struct PlyProp { std::string name; bool list = false; PlyType type = P_BAD, count_type = P_BAD; };
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };

// ascii and binary_little_endian; vertex x y z (+ s t or u v), face vertex_indices / vertex_index
bool load_ply(const char *path, uint32_t material, std::vector<kl_tri> &out, std::string &err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = std::string("cannot read ") + path; return false; }
    std::string line, kw;
    if (!std::getline(in, line) || line.compare(0, 3, "ply") != 0) { err = "not a PLY file"; return false; }
    bool ascii = true, have_format = false;
    std::vector<PlyElem> elems;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        if (!(ls >> kw)) continue;
        if (kw == "format") {
            std::string f; ls >> f; have_format = true;
            if (f == "ascii") ascii = true;
            else if (f == "binary_little_endian") ascii = false;
            else { err = "PLY format '" + f + "' is not supported (ascii and binary_little_endian are)"; return false; }
        } else if (kw == "element") {
            PlyElem e; long long cnt = -1;
            if (!(ls >> e.name >> cnt) || cnt < 0) { err = "bad PLY element line: " + line; return false; }
            e.count = (size_t)cnt; elems.push_back(e);
        } else if (kw == "property") {
            if (elems.empty()) { err = "PLY property before any element"; return false; }
            PlyProp p; std::string t; ls >> t;
            if (t == "list") { std::string ct, it; ls >> ct >> it >> p.name; p.list = true; p.count_type = ply_type(ct); p.type = ply_type(it); }
            else { p.type = ply_type(t); ls >> p.name; }
            if (p.type == P_BAD || (p.list && p.count_type == P_BAD)) { err = "PLY property of unknown type in: " + line; return false; }
            elems.back().props.push_back(p);
        } else if (kw == "end_header") break;
    }
    if (!have_format) { err = "PLY header has no format line"; return false; }
    // Header counts are claims of the file: every element instance takes at least one byte per property (ascii: a
    // digit and a separator share two), so a count beyond the bytes that remain, or an element without properties,
    // cannot be honest. This also bounds the memory the loops below can ask for.
    {
        const std::streampos here = in.tellg();
        in.seekg(0, std::ios::end);
        const std::streampos end = in.tellg();
        in.seekg(here);
        const unsigned long long remain = (here >= 0 && end >= here) ? (unsigned long long)(end - here) : 0ull;
        for (const PlyElem &e : elems) {
            if (e.count == 0) continue;
            if (e.props.empty()) { err = "PLY element '" + e.name + "' has a count but no properties"; return false; }
            if ((unsigned long long)e.count > remain) { err = "PLY element count exceeds the size of the file"; return false; }
            if ((e.name == "vertex" || e.name == "face") && e.count >= kMaxElements) { err = "PLY element count beyond 2^27"; return false; }
        }
    }
    std::vector<V3> v; std::vector<V2> vt;
    auto next_scalar = [&](PlyType t, double &val) -> bool {
        if (ascii) return (bool)(in >> val);
        unsigned char buf[8];
        if (!in.read((char *)buf, (std::streamsize)ply_size(t))) return false;
        val = ply_value(buf, t); return true;
    };
    for (const PlyElem &e : elems) {
        for (size_t i = 0; i < e.count; i++) {
            V3 p{0, 0, 0}; V2 t{0, 0}; bool has_uv = false;
            std::vector<long> idx;
            for (const PlyProp &pr : e.props) {
                double val = 0.0;
                if (pr.list) {
                    if (!next_scalar(pr.count_type, val) || val < 0 || val > 1e6) { err = "truncated PLY data"; return false; }
                    const size_t n = (size_t)val;
                    const bool is_index = e.name == "face" && (pr.name == "vertex_indices" || pr.name == "vertex_index");
                    for (size_t k = 0; k < n; k++) {
                        if (!next_scalar(pr.type, val)) { err = "truncated PLY data"; return false; }
                        if (is_index) {
                            // an index that is not a finite value inside the vertex array is rejected before the cast
                            if (!(val >= 0.0 && val < (double)kMaxElements)) { err = "PLY face index out of range"; return false; }
                            idx.push_back((long)val);
                        }
                    }
                } else {
                    if (!next_scalar(pr.type, val)) { err = "truncated PLY data"; return false; }
                    if (e.name == "vertex") {
                        if (pr.name == "x") p.x = (float)val; else if (pr.name == "y") p.y = (float)val; else if (pr.name == "z") p.z = (float)val;
                        else if (pr.name == "s" || pr.name == "u" || pr.name == "texture_u") { t.u = (float)val; has_uv = true; }
                        else if (pr.name == "t" || pr.name == "v" || pr.name == "texture_v") { t.v = (float)val; has_uv = true; }
                    }
                }
            }
            if (e.name == "vertex") { v.push_back(p); vt.push_back(has_uv ? t : V2{0, 0}); }
            else if (e.name == "face") {
                if (idx.size() < 3) { err = "PLY face with fewer than 3 corners"; return false; }
                for (long k : idx) if (k < 0 || (size_t)k >= v.size()) { err = "PLY face index out of range"; return false; }
                if (out.size() + idx.size() > kMaxTriangles) { err = "more than 2^25 triangles"; return false; }
                for (size_t k = 1; k + 1 < idx.size(); k++) {
                    kl_tri tr;
                    put(tr, v[idx[0]], v[idx[k]], v[idx[k + 1]], vt[idx[0]], vt[idx[k]], vt[idx[k + 1]], material);
                    out.push_back(tr);
                }
            }
        }
    }
    return true;
}

thread_local std::string g_mesh_error;

}  // namespace

extern "C" {

static kl_status load_mesh_body(const char *path, uint32_t material, uint32_t n_materials, kl_tri **tris, uint32_t *n_tris) {
    std::vector<kl_tri> out;
    const std::string p(path);
    const size_t dot = p.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot + 1);
    for (char &ch : ext) ch = (char)std::tolower((unsigned char)ch);
    bool ok;
    g_mesh_error.clear();
    if (ext == "obj") ok = load_obj(path, material, n_materials, out, g_mesh_error);
    else if (ext == "ply") ok = load_ply(path, material, out, g_mesh_error);
    else { g_mesh_error = "unknown mesh file extension (.obj and .ply are read)"; return KL_E_INVALID; }
    if (!ok) return KL_E_INVALID;
    if (out.size() >= (1u << 25)) { g_mesh_error = "more than 2^25 triangles"; return KL_E_LIMIT; }
    if (!out.empty()) {
        kl_tri *buf = (kl_tri *)std::malloc(out.size() * sizeof(kl_tri));
        if (!buf) return KL_E_NOMEM;
        std::memcpy(buf, out.data(), out.size() * sizeof(kl_tri));
        *tris = buf;
    }
    *n_tris = (uint32_t)out.size();
    return KL_OK;
}

kl_status kl_load_mesh(const char *path, uint32_t material, uint32_t n_materials, kl_tri **tris, uint32_t *n_tris) {
    if (!path || !tris || !n_tris) return KL_E_INVALID;
    *tris = nullptr; *n_tris = 0;
    // no exception crosses the C ABI (the loaders grow std::vectors from what a file claims)
    try { return load_mesh_body(path, material, n_materials, tris, n_tris); }
    catch (const std::bad_alloc &) { try { g_mesh_error = "host memory exhausted"; } catch (...) {} return KL_E_NOMEM; }
    catch (const std::exception &e) { try { g_mesh_error = e.what(); } catch (...) {} return KL_E_INVALID; }
    catch (...) { return KL_E_INVALID; }
}

const char *kl_load_mesh_error(void) { return g_mesh_error.c_str(); }

void kl_free(void *p) { std::free(p); }

}  // extern "C"
