// Binned-SAH BVH builder (host, multi-threaded). See kl_bvh.hpp for the guarantees.
#include "kl_bvh.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>

#include "kl_types.hpp"

namespace kl {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; } }
    void grow(const Box &b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float p[3]) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct Prim { Box box; float c[3]; uint32_t type, index; };

struct TmpNode {
    Box box[2];
    int32_t child[2];     // >= 0: TmpNode index; < 0: ~leaf index
};
struct TmpLeaf { uint32_t type, start, count; };   // range in Builder::prims

constexpr int MAX_BINS = 64;
static int NBINS = 16;
// tuning knobs of the builder (environment overrides are for experiments): runs of at most leaf_target
// primitives become a leaf without asking the SAH; trav_cost = cost of one binary node visit relative to one primitive test
static uint32_t LEAF_TARGET = 2;
static float TRAV_COST = 1.0f;

struct Builder {
    std::vector<Prim> prims;
    std::vector<TmpNode> nodes;
    std::vector<TmpLeaf> leaves;
    std::atomic<uint32_t> n_nodes{0}, n_leaves{0};
    std::atomic<int> tasks{0};
    int max_tasks = 1;

    static uint32_t ceil_log2(uint32_t n) { uint32_t l = 0; while ((1u << l) < n) l++; return l; }

    int32_t make_leaf(uint32_t b, uint32_t e) {
        uint32_t li = n_leaves.fetch_add(1);
        leaves[li].type = prims[b].type; leaves[li].start = b; leaves[li].count = e - b;
        return ~(int32_t)li;
    }
    bool single_type(uint32_t b, uint32_t e) const {
        for (uint32_t i = b + 1; i < e; i++) if (prims[i].type != prims[b].type) return false;
        return true;
    }
    Box bounds(uint32_t b, uint32_t e) const { Box x; x.reset(); for (uint32_t i = b; i < e; i++) x.grow(prims[i].box); return x; }

    // returns the child reference for prims [b, e); `depth` = number of inner nodes above
    int32_t build(uint32_t b, uint32_t e, int depth) {
        const uint32_t n = e - b;
        if (n == 1) return make_leaf(b, e);
        const bool homo = single_type(b, e);
        // (triangles only: two neighbours of a mesh share an edge and a tight box; two spheres or quads that merely lie next to
        // each other do not, and the SAH below decides about them -- 2.08 -> 1.39 sphere tests per ray on the mixed 10 M scene)
        if (n <= (prims[b].type == PRIM_TRI ? LEAF_TARGET : 1u) && homo) return make_leaf(b, e);
        uint32_t mid = 0;
        bool have_split = false;
        if (n <= LEAF_MAX && !homo) {
            // small mixed run: separate the types (at most two such levels)
            uint32_t t0 = prims[b].type;
            for (uint32_t i = b; i < e; i++) t0 = std::min(t0, prims[i].type);
            auto it = std::partition(prims.begin() + b, prims.begin() + e, [t0](const Prim &p) { return p.type == t0; });
            mid = (uint32_t)(it - prims.begin()); have_split = true;
        }
        const int rem = BVH_MAX_DEPTH - depth;
        const bool forced_median = (int)ceil_log2(n) + 2 >= rem;   // out of depth: balanced splits only from here down
        if (forced_median && n <= LEAF_MAX && homo) return make_leaf(b, e);
        if (!have_split) {
            Box cb; cb.reset();
            for (uint32_t i = b; i < e; i++) cb.grow(prims[i].c);
            int axis = 0;
            float ext[3] = { cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2] };
            if (ext[1] > ext[axis]) axis = 1;
            if (ext[2] > ext[axis]) axis = 2;
            if (!forced_median && ext[axis] > 0.0f) {
                float best_cost = std::numeric_limits<float>::infinity(); int best_axis = -1, best_bin = -1;
                for (int ax = 0; ax < 3; ax++) {
                    if (!(ext[ax] > 0.0f)) continue;
                    Box bb[MAX_BINS]; uint32_t bn[MAX_BINS];
                    for (int k = 0; k < NBINS; k++) { bb[k].reset(); bn[k] = 0; }
                    const float scale = (float)NBINS / ext[ax];
                    for (uint32_t i = b; i < e; i++) {
                        int k = (int)((prims[i].c[ax] - cb.lo[ax]) * scale);
                        k = k < 0 ? 0 : (k >= NBINS ? NBINS - 1 : k);
                        bb[k].grow(prims[i].box); bn[k]++;
                    }
                    float right_area[MAX_BINS]; uint32_t right_n[MAX_BINS];
                    Box acc; acc.reset(); uint32_t cnt = 0;
                    for (int k = NBINS - 1; k > 0; k--) { acc.grow(bb[k]); cnt += bn[k]; right_area[k] = acc.area(); right_n[k] = cnt; }
                    acc.reset(); cnt = 0;
                    for (int k = 0; k < NBINS - 1; k++) {
                        acc.grow(bb[k]); cnt += bn[k];
                        if (cnt == 0 || right_n[k + 1] == 0) continue;
                        float cost = acc.area() * (float)cnt + right_area[k + 1] * (float)right_n[k + 1];
                        if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
                    }
                }
                if (best_axis >= 0) {
                    if (n <= LEAF_MAX && homo) {
                        // SAH termination: leaf cost n*A against one node visit plus the children
                        float pa = bounds(b, e).area();
                        if ((float)n * pa <= TRAV_COST * pa + best_cost) return make_leaf(b, e);
                    }
                    const float scale = (float)NBINS / ext[best_axis];
                    const float lo = cb.lo[best_axis];
                    const int bax = best_axis, bbin = best_bin;
                    auto it = std::partition(prims.begin() + b, prims.begin() + e, [=](const Prim &p) {
                        int k = (int)((p.c[bax] - lo) * scale);
                        k = k < 0 ? 0 : (k >= NBINS ? NBINS - 1 : k);
                        return k <= bbin;
                    });
                    mid = (uint32_t)(it - prims.begin());
                    have_split = mid > b && mid < e;
                }
            }
            if (!have_split) {
                if (n <= LEAF_MAX && homo) return make_leaf(b, e);   // coincident centroids
                mid = b + n / 2;
                std::nth_element(prims.begin() + b, prims.begin() + mid, prims.begin() + e,
                                 [axis](const Prim &x, const Prim &y) { return x.c[axis] < y.c[axis]; });
            }
        }
        uint32_t ni = n_nodes.fetch_add(1);
        int32_t c0, c1;
        if (n > 32768 && tasks.load() < max_tasks) {
            tasks.fetch_add(1);
            auto fut = std::async(std::launch::async, [this, b, mid, depth]() { return build(b, mid, depth + 1); });
            c1 = build(mid, e, depth + 1);
            c0 = fut.get();
            tasks.fetch_sub(1);
        } else {
            c0 = build(b, mid, depth + 1);
            c1 = build(mid, e, depth + 1);
        }
        nodes[ni].child[0] = c0; nodes[ni].child[1] = c1;
        nodes[ni].box[0] = bounds(b, mid); nodes[ni].box[1] = bounds(mid, e);
        return (int32_t)ni;
    }
};

void prim_box(const float *const *v, int nv, Box &b) {
    b.reset();
    for (int i = 0; i < nv; i++) b.grow(v[i]);
}
void pad_box(Box &b) {
    for (int k = 0; k < 3; k++) {
        float m = std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k]));
        float pad = 1e-6f + 4e-6f * m;
        b.lo[k] -= pad; b.hi[k] += pad;
    }
}

}  // namespace

namespace {
struct Child4 { Box box; int32_t ref; };

// Fills 4-wide node `idx` (already allocated in `out`) from binary node `node2`. The inner children of a node
// get consecutive indices ("siblings side by side"): a ray that enters one child usually comes back for its
// neighbours a few steps later, and two 64-byte siblings share a 128-byte cache line.
void collapse_node(const HostBVH &bvh, int32_t node2, uint32_t idx, std::vector<uint32_t> &out, uint32_t depth, uint32_t &max_depth) {
    auto child_of = [&](int32_t n2, int k) {
        const float *o = &bvh.nodes[(size_t)n2 * 16];
        Child4 c;
        for (int a = 0; a < 3; a++) { c.box.lo[a] = o[6 * k + a]; c.box.hi[a] = o[6 * k + 3 + a]; }
        std::memcpy(&c.ref, o + 12 + k, 4);
        return c;
    };
    std::vector<Child4> ch = { child_of(node2, 0), child_of(node2, 1) };
    while (ch.size() < 4) {            // open the inner child with the largest box
        int best = -1; float best_area = -1.0f;
        for (size_t i = 0; i < ch.size(); i++)
            if (ch[i].ref >= 0 && ch[i].box.area() > best_area) { best = (int)i; best_area = ch[i].box.area(); }
        if (best < 0) break;
        int32_t open = ch[best].ref;
        ch[best] = child_of(open, 0);
        ch.push_back(child_of(open, 1));
    }
    max_depth = std::max(max_depth, depth);
    uint32_t w[16]; std::memset(w, 0, sizeof w);
    float boxes[4][6];
    for (size_t k = 0; k < ch.size(); k++)
        for (int a = 0; a < 3; a++) { boxes[k][a] = ch[k].box.lo[a]; boxes[k][3 + a] = ch[k].box.hi[a]; }
    node4_boxes(&boxes[0][0], (int)ch.size(), w);
    uint32_t next = (uint32_t)(out.size() / 16);           // inner children: consecutive indices from here
    uint32_t n_inner = 0;
    for (const Child4 &c : ch) n_inner += c.ref >= 0 ? 1u : 0u;
    out.resize(out.size() + (size_t)16 * n_inner, 0u);
    for (int k = 0; k < 4; k++) {
        int32_t ref = NODE_DONE;
        if (k < (int)ch.size()) ref = ch[k].ref >= 0 ? (int32_t)next++ : ch[k].ref;
        std::memcpy(&w[12 + k], &ref, 4);
    }
    // an empty slot repeats child 0: always a valid reference (see kl_types.hpp)
    for (int k = (int)ch.size(); k < 4; k++) w[12 + k] = w[12];
    std::memcpy(&out[(size_t)idx * 16], w, sizeof w);
    for (int k = 0; k < (int)ch.size(); k++)
        if (ch[k].ref >= 0) collapse_node(bvh, ch[k].ref, w[12 + k], out, depth + 1, max_depth);
}

// ---- collapse by dynamic programming over the binary tree --------------------------------------------------
// Cost of a traversal = expected steps: one per 4-wide node entered, one per primitive of a leaf entered, each
// weighted by the area of the box that guards it. For a binary node n and k = 1..4, f(n, k) = the cheapest way
// to hang n's subtree under a 4-wide parent using at most k of the parent's slots: as ONE child (a 4-wide node
// of its own: area + the best split of its two children over four slots; or, when its two children are leaves
// of one type next to each other in the leaf order, one merged leaf: area x primitives), or with its two
// children sharing the k slots. The greedy rule (open the largest child) is one of the shapes considered.
inline uint32_t leaf_type(int32_t r) { return ((uint32_t)~r >> 29) & 3u; }
inline uint32_t leaf_count(int32_t r) { return (((uint32_t)~r >> 26) & 7u) + 1u; }
inline uint32_t leaf_start(int32_t r) { return (uint32_t)~r & 0x3FFFFFFu; }

struct DpCollapse {
    const HostBVH &bvh;
    std::vector<float> f;            // 4 per binary node: f[4 n + k - 1]
    std::vector<int32_t> merged;     // the merged leaf when that is n's best form as one child, else 0
    std::vector<uint32_t> out;
    uint32_t max_depth = 0, leaves = 0, max_leaf = 0;

    explicit DpCollapse(const HostBVH &b) : bvh(b), f(b.nodes.size() / 16 * 4), merged(b.nodes.size() / 16, 0) {}
    Child4 child_of(int32_t n2, int k) const {
        const float *o = &bvh.nodes[(size_t)n2 * 16];
        Child4 c;
        for (int a = 0; a < 3; a++) { c.box.lo[a] = o[6 * k + a]; c.box.hi[a] = o[6 * k + 3 + a]; }
        std::memcpy(&c.ref, o + 12 + k, 4);
        return c;
    }
    float cost(const Child4 &c, int k) const { return c.ref < 0 ? c.box.area() * (float)leaf_count(c.ref) : f[(size_t)c.ref * 4 + k - 1]; }
    // cheapest split of k slots between the two children; returns the cost, `left` = slots of child 0
    float split(const Child4 &c0, const Child4 &c1, int k, int &left) const {
        float best = std::numeric_limits<float>::infinity();
        left = 1;
        for (int i = 1; i < k; i++) {
            const float c = cost(c0, i) + cost(c1, k - i);
            if (c < best) { best = c; left = i; }
        }
        return best;
    }
    void solve(int32_t root) {       // children before parents, without recursion (a binary tree may be deep)
        std::vector<std::pair<int32_t, bool>> stack = { { root, false } };
        while (!stack.empty()) {
            auto [n, ready] = stack.back(); stack.pop_back();
            const Child4 c0 = child_of(n, 0), c1 = child_of(n, 1);
            if (!ready) {
                stack.push_back({ n, true });
                if (c0.ref >= 0) stack.push_back({ c0.ref, false });
                if (c1.ref >= 0) stack.push_back({ c1.ref, false });
                continue;
            }
            Box nb = c0.box; nb.grow(c1.box);
            const float area = nb.area();
            int left;
            float one = area + split(c0, c1, 4, left);
            if (c0.ref < 0 && c1.ref < 0 && leaf_type(c0.ref) == leaf_type(c1.ref) && leaf_start(c1.ref) == leaf_start(c0.ref) + leaf_count(c0.ref)
                && leaf_count(c0.ref) + leaf_count(c1.ref) <= LEAF_MAX) {
                const float as_leaf = area * (float)(leaf_count(c0.ref) + leaf_count(c1.ref));
                if (as_leaf <= one) { one = as_leaf; merged[n] = make_leaf(leaf_type(c0.ref), leaf_start(c0.ref), leaf_count(c0.ref) + leaf_count(c1.ref)); }
            }
            float *fn = &f[(size_t)n * 4];
            fn[0] = one;
            for (int k = 2; k <= 4; k++) fn[k - 1] = std::min(fn[k - 2], split(c0, c1, k, left));
        }
    }
    // the children that `c` contributes to a 4-wide node when it may use up to k slots
    void gather(const Child4 &c, int k, std::vector<Child4> &ch) const {
        if (c.ref < 0) { ch.push_back(c); return; }
        const Child4 c0 = child_of(c.ref, 0), c1 = child_of(c.ref, 1);
        const float *fn = &f[(size_t)c.ref * 4];
        while (k > 1) {
            int left;
            if (split(c0, c1, k, left) < fn[k - 2]) { gather(c0, left, ch); gather(c1, k - left, ch); return; }
            k--;
        }
        Child4 one = c;
        if (merged[c.ref]) one.ref = merged[c.ref];
        ch.push_back(one);
    }
    void emit(int32_t node2, uint32_t idx, uint32_t depth) {
        const Child4 c0 = child_of(node2, 0), c1 = child_of(node2, 1);
        int left;
        split(c0, c1, 4, left);
        std::vector<Child4> ch;
        gather(c0, left, ch); gather(c1, 4 - left, ch);
        max_depth = std::max(max_depth, depth);
        uint32_t w[16]; std::memset(w, 0, sizeof w);
        float boxes[4][6];
        for (size_t k = 0; k < ch.size(); k++)
            for (int a = 0; a < 3; a++) { boxes[k][a] = ch[k].box.lo[a]; boxes[k][3 + a] = ch[k].box.hi[a]; }
        node4_boxes(&boxes[0][0], (int)ch.size(), w);
        uint32_t next = (uint32_t)(out.size() / 16);           // inner children: consecutive indices from here
        uint32_t n_inner = 0;
        for (const Child4 &c : ch) n_inner += c.ref >= 0 ? 1u : 0u;
        out.resize(out.size() + (size_t)16 * n_inner, 0u);
        for (size_t k = 0; k < ch.size(); k++) {
            if (ch[k].ref >= 0) w[12 + k] = next++;
            else { w[12 + k] = (uint32_t)ch[k].ref; leaves++; max_leaf = std::max(max_leaf, leaf_count(ch[k].ref)); }
        }
        for (size_t k = ch.size(); k < 4; k++) w[12 + k] = w[12];      // an empty slot repeats child 0 (see kl_types.hpp)
        std::memcpy(&out[(size_t)idx * 16], w, sizeof w);
        for (size_t k = 0; k < ch.size(); k++)
            if (ch[k].ref >= 0) emit(ch[k].ref, w[12 + k], depth + 1);
    }
};
}  // namespace

void collapse_bvh4(HostBVH &bvh) {
    bvh.nodes4.clear(); bvh.depth4 = 0;
    bvh.root4 = bvh.root;                       // an empty tree or a lone leaf stays what it is
    if (bvh.root < 0 || bvh.root == NODE_DONE) return;
    const char *how = std::getenv("KL_BVH_COLLAPSE");
    if (how && std::strcmp(how, "greedy") == 0) {
        bvh.nodes4.reserve(bvh.nodes.size() / 2);
        bvh.nodes4.assign(16, 0u);                // the root is node 0
        collapse_node(bvh, bvh.root, 0u, bvh.nodes4, 1, bvh.depth4);
        bvh.root4 = 0;
        return;
    }
    DpCollapse dp(bvh);
    dp.solve(bvh.root);
    if (dp.merged[bvh.root]) {                  // the whole tree is one leaf
        bvh.root4 = dp.merged[bvh.root];
        bvh.stats.leaves = 1; bvh.stats.max_leaf_size = leaf_count(bvh.root4);
        return;
    }
    dp.out.reserve(bvh.nodes.size() / 2);
    dp.out.assign(16, 0u);
    dp.emit(bvh.root, 0u, 1);
    bvh.nodes4.swap(dp.out);
    bvh.depth4 = dp.max_depth; bvh.root4 = 0;
    bvh.stats.leaves = dp.leaves; bvh.stats.max_leaf_size = dp.max_leaf;
}

void identity_order(uint32_t ns, uint32_t nq, uint32_t nt, HostBVH &out) {
    out.nodes.clear(); out.root = NODE_DONE;
    out.sph_order.resize(ns); out.quad_order.resize(nq); out.tri_order.resize(nt);
    for (uint32_t i = 0; i < ns; i++) out.sph_order[i] = i;
    for (uint32_t i = 0; i < nq; i++) out.quad_order[i] = i;
    for (uint32_t i = 0; i < nt; i++) out.tri_order[i] = i;
    std::memset(&out.stats, 0, sizeof out.stats);
    out.stats.spheres = ns; out.stats.quads = nq; out.stats.tris = nt;
    out.flat_start[PRIM_SPHERE] = ns; out.flat_start[PRIM_QUAD] = nq; out.flat_start[PRIM_TRI] = nt;
    out.flat_count[0] = out.flat_count[1] = out.flat_count[2] = 0;
}

void build_bvh(const kl_sphere *s, uint32_t ns, const kl_quad *q, uint32_t nq, const kl_tri *t, uint32_t nt,
               HostBVH &out, unsigned threads, float flat_fraction, uint32_t max_flat) {
    auto t0 = std::chrono::steady_clock::now();
    if (const char *e = std::getenv("KL_BVH_LEAF_TARGET")) LEAF_TARGET = (uint32_t)std::max(1, std::min(8, std::atoi(e)));
    if (const char *e = std::getenv("KL_BVH_TRAV_COST")) TRAV_COST = (float)std::atof(e);
    if (const char *e = std::getenv("KL_BVH_BINS")) NBINS = std::max(4, std::min(MAX_BINS, std::atoi(e)));
    uint32_t n = ns + nq + nt;
    identity_order(0, 0, 0, out);
    out.stats.spheres = ns; out.stats.quads = nq; out.stats.tris = nt;
    if (n == 0) return;
    Builder B;
    B.prims.resize(n);
    for (uint32_t i = 0; i < ns; i++) {
        Prim &p = B.prims[i]; p.type = PRIM_SPHERE; p.index = i;
        for (int k = 0; k < 3; k++) { p.box.lo[k] = s[i].center[k] - s[i].radius; p.box.hi[k] = s[i].center[k] + s[i].radius; }
    }
    for (uint32_t i = 0; i < nq; i++) {
        Prim &p = B.prims[ns + i]; p.type = PRIM_QUAD; p.index = i;
        const float *v[4] = { q[i].v0, q[i].v1, q[i].v2, q[i].v3 };
        prim_box(v, 4, p.box);
    }
    for (uint32_t i = 0; i < nt; i++) {
        Prim &p = B.prims[ns + nq + i]; p.type = PRIM_TRI; p.index = i;
        const float *v[3] = { t[i].v0, t[i].v1, t[i].v2 };
        prim_box(v, 3, p.box);
    }
    for (Prim &p : B.prims) {
        pad_box(p.box);
        for (int k = 0; k < 3; k++) p.c[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
    }
    // split off the flat primitives: [0, n) stays in the tree, [n, n_all) is flat
    const uint32_t n_all = n;
    if (flat_fraction > 0.0f && max_flat > 0) {
        const float limit = flat_fraction * B.bounds(0, n_all).area();
        auto mid = std::partition(B.prims.begin(), B.prims.end(), [limit](const Prim &p) { return !(p.box.area() >= limit); });
        uint32_t first_flat = (uint32_t)(mid - B.prims.begin());
        if (n_all - first_flat > max_flat) {      // keep only the largest max_flat; the others rejoin the tree
            std::sort(B.prims.begin() + first_flat, B.prims.end(), [](const Prim &a, const Prim &b) { return a.box.area() < b.box.area(); });
            first_flat = n_all - max_flat;
        }
        n = first_flat;
    }
    B.nodes.resize(n + 1); B.leaves.resize(n + 1);
    B.max_tasks = threads > 1 ? (int)threads - 1 : 0;
    int32_t root = n ? B.build(0, n, 0) : 0;

    // flatten depth-first; leaves append their primitives to the per-type leaf order
    const uint32_t nn = B.n_nodes.load();
    out.nodes.assign((size_t)nn * 16, 0.0f);
    out.sph_order.reserve(ns); out.quad_order.reserve(nq); out.tri_order.reserve(nt);
    uint32_t next = 0, max_depth = 0, max_leaf = 0, n_leaves = 0;
    double sah = 0.0;
    struct Item { int32_t tmp; int32_t *slot; uint32_t depth; };
    std::vector<Item> stack;
    int32_t root_ref = NODE_DONE;
    if (n) stack.push_back({root, &root_ref, 0});
    std::vector<int32_t> child_store((size_t)nn * 2);
    Box rootbox = B.bounds(0, n ? n : n_all);
    const float root_area = std::max(rootbox.area(), 1e-30f);
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        if (it.tmp < 0) {
            const TmpLeaf &lf = B.leaves[(uint32_t)~it.tmp];
            std::vector<uint32_t> &ord = lf.type == PRIM_SPHERE ? out.sph_order : (lf.type == PRIM_QUAD ? out.quad_order : out.tri_order);
            uint32_t start = (uint32_t)ord.size();
            for (uint32_t i = 0; i < lf.count; i++) ord.push_back(B.prims[lf.start + i].index);
            *it.slot = make_leaf(lf.type, start, lf.count);
            max_depth = std::max(max_depth, it.depth); max_leaf = std::max(max_leaf, lf.count); n_leaves++;
            sah += (double)B.bounds(lf.start, lf.start + lf.count).area() / root_area * lf.count;
        } else {
            const TmpNode &tn = B.nodes[(uint32_t)it.tmp];
            uint32_t idx = next++;
            *it.slot = (int32_t)idx;
            float *o = &out.nodes[(size_t)idx * 16];
            o[0] = tn.box[0].lo[0]; o[1] = tn.box[0].lo[1]; o[2] = tn.box[0].lo[2];
            o[3] = tn.box[0].hi[0]; o[4] = tn.box[0].hi[1]; o[5] = tn.box[0].hi[2];
            o[6] = tn.box[1].lo[0]; o[7] = tn.box[1].lo[1]; o[8] = tn.box[1].lo[2];
            o[9] = tn.box[1].hi[0]; o[10] = tn.box[1].hi[1]; o[11] = tn.box[1].hi[2];
            sah += (double)(tn.box[0].area() + tn.box[1].area()) / root_area * 0.5 * 2.0 * 0.5;  // one visit per entered child box
            // children are patched in after they get their final index; child 0 is laid out next
            stack.push_back({tn.child[1], &child_store[(size_t)idx * 2 + 1], it.depth + 1});
            stack.push_back({tn.child[0], &child_store[(size_t)idx * 2 + 0], it.depth + 1});
        }
    }
    for (uint32_t i = 0; i < nn; i++) {
        std::memcpy(&out.nodes[(size_t)i * 16 + 12], &child_store[(size_t)i * 2], 4);
        std::memcpy(&out.nodes[(size_t)i * 16 + 13], &child_store[(size_t)i * 2 + 1], 4);
    }
    // flat primitives close each order array
    out.flat_start[PRIM_SPHERE] = (uint32_t)out.sph_order.size();
    out.flat_start[PRIM_QUAD] = (uint32_t)out.quad_order.size();
    out.flat_start[PRIM_TRI] = (uint32_t)out.tri_order.size();
    for (uint32_t i = n; i < n_all; i++) {
        const Prim &p = B.prims[i];
        (p.type == PRIM_SPHERE ? out.sph_order : (p.type == PRIM_QUAD ? out.quad_order : out.tri_order)).push_back(p.index);
        out.flat_count[p.type]++;
    }
    out.root = root_ref;
    out.stats.nodes = nn; out.stats.leaves = n_leaves; out.stats.max_depth = max_depth; out.stats.max_leaf_size = max_leaf;
    out.stats.sah_cost = (float)sah;
    out.stats.node_bytes = (uint64_t)nn * 64u;
    out.stats.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace kl
