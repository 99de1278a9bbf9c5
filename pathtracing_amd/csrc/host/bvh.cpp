#include "bvh.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <future>
#include <limits>

namespace pt {
namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr int kBins = 16;
constexpr int kSahDepthLimit = 30; // below this depth only balanced median splits

struct Ref {
    float bmin[3], bmax[3], c[3];
    uint32_t idx;
};

struct Box {
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    void grow(const float* a, const float* b) {
        for (int k = 0; k < 3; k++) {
            mn[k] = std::min(mn[k], a[k]);
            mx[k] = std::max(mx[k], b[k]);
        }
    }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct SubTree {
    std::vector<ptk::BvhNode> nodes; // child refs >= 0 are local to this vector
    std::vector<uint32_t> order;     // leaf refs index into this vector
    uint32_t max_depth = 0;
};

struct Builder {
    std::vector<Ref>& refs;
    uint32_t max_leaf;

    // Chooses where to split refs[lo, hi); returns the split position.
    size_t split(size_t lo, size_t hi, int depth) {
        size_t n = hi - lo;
        Box cb;
        for (size_t i = lo; i < hi; i++) cb.grow(refs[i].c, refs[i].c);
        int best_axis = -1, best_bin = -1;
        float best_cost = kInf;
        if (depth < kSahDepthLimit) {
            for (int a = 0; a < 3; a++) {
                float ext = cb.mx[a] - cb.mn[a];
                if (!(ext > 0.0f)) continue;
                Box bins[kBins];
                uint32_t cnt[kBins] = {0};
                float scale = (float)kBins / ext;
                for (size_t i = lo; i < hi; i++) {
                    int b = std::min(kBins - 1, std::max(0, (int)((refs[i].c[a] - cb.mn[a]) * scale)));
                    cnt[b]++;
                    bins[b].grow(refs[i].bmin, refs[i].bmax);
                }
                float right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                uint32_t c = 0;
                for (int b = kBins - 1; b >= 1; b--) {
                    if (cnt[b]) acc.grow(bins[b].mn, bins[b].mx);
                    c += cnt[b];
                    right_area[b] = c ? acc.half_area() : 0.0f;
                    right_cnt[b] = c;
                }
                acc = Box();
                c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    if (cnt[b]) acc.grow(bins[b].mn, bins[b].mx);
                    c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_bin = b;
                    }
                }
            }
        }
        if (best_axis >= 0) {
            float ext = cb.mx[best_axis] - cb.mn[best_axis];
            float scale = (float)kBins / ext, base = cb.mn[best_axis];
            int a = best_axis, sb = best_bin;
            auto it = std::partition(refs.begin() + lo, refs.begin() + hi, [=](const Ref& r) {
                int b = std::min(kBins - 1, std::max(0, (int)((r.c[a] - base) * scale)));
                return b <= sb;
            });
            size_t mid = (size_t)(it - refs.begin());
            if (mid > lo && mid < hi) return mid;
        }
        // balanced split on the widest centroid axis (ties broken by index: deterministic)
        int a = 0;
        float e = -1.0f;
        for (int k = 0; k < 3; k++)
            if (cb.mx[k] - cb.mn[k] > e) {
                e = cb.mx[k] - cb.mn[k];
                a = k;
            }
        size_t mid = lo + n / 2;
        std::nth_element(refs.begin() + lo, refs.begin() + mid, refs.begin() + hi, [a](const Ref& p, const Ref& q) {
            return p.c[a] < q.c[a] || (p.c[a] == q.c[a] && p.idx < q.idx);
        });
        return mid;
    }

    // Builds refs[lo, hi) into `out`; returns the child reference and its bounds.
    int32_t build(SubTree& out, size_t lo, size_t hi, int depth, Box* bounds) {
        size_t n = hi - lo;
        Box b;
        for (size_t i = lo; i < hi; i++) b.grow(refs[i].bmin, refs[i].bmax);
        *bounds = b;
        out.max_depth = std::max(out.max_depth, (uint32_t)depth);
        if (n <= max_leaf) {
            uint32_t first = (uint32_t)out.order.size();
            for (size_t i = lo; i < hi; i++) out.order.push_back(refs[i].idx);
            return ptk::bvh_make_leaf(first, (uint32_t)n);
        }
        size_t mid = split(lo, hi, depth);
        int32_t me = (int32_t)out.nodes.size();
        out.nodes.emplace_back();
        Box lb, rb;
        int32_t l = build(out, lo, mid, depth + 1, &lb);
        int32_t r = build(out, mid, hi, depth + 1, &rb);
        ptk::BvhNode& nd = out.nodes[me];
        for (int k = 0; k < 3; k++) {
            nd.lmin[k] = lb.mn[k];
            nd.lmax[k] = lb.mx[k];
            nd.rmin[k] = rb.mn[k];
            nd.rmax[k] = rb.mx[k];
        }
        nd.left = l;
        nd.right = r;
        nd.pad[0] = nd.pad[1] = 0;
        return me;
    }
};

// Appends `sub` to `dst`, rebasing node indices and leaf offsets; returns the
// rebased reference of sub's root (`root`).
int32_t append_subtree(SubTree& dst, const SubTree& sub, int32_t root) {
    int32_t node_base = (int32_t)dst.nodes.size();
    uint32_t tri_base = (uint32_t)dst.order.size();
    auto rebase = [&](int32_t ref) -> int32_t {
        if (ref >= 0) return ref + node_base;
        uint32_t raw = (uint32_t)(~ref);
        return ptk::bvh_make_leaf((raw >> 3) + tri_base, (raw & 7u) + 1u);
    };
    for (ptk::BvhNode nd : sub.nodes) {
        nd.left = rebase(nd.left);
        nd.right = rebase(nd.right);
        dst.nodes.push_back(nd);
    }
    dst.order.insert(dst.order.end(), sub.order.begin(), sub.order.end());
    dst.max_depth = std::max(dst.max_depth, sub.max_depth);
    return rebase(root);
}

uint64_t fnv1a(const void* data, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

struct TreeHeader {
    char magic[8]; // "PTAMDTRE"
    uint32_t version;
    uint32_t node_size;
    uint64_t triangle_count;
    uint64_t node_count;
    int32_t root_ref;
    uint32_t max_depth;
    float bmin[3], bmax[3];
    uint64_t geometry_hash; // FNV-1a of the position floats
};
constexpr uint32_t kTreeVersion = 1;

} // namespace

float padding_reach(const float* mesh_bmin, const float* mesh_bmax) {
    float dx = mesh_bmax[0] - mesh_bmin[0], dy = mesh_bmax[1] - mesh_bmin[1], dz = mesh_bmax[2] - mesh_bmin[2];
    return 4.0f * std::sqrt(dx * dx + dy * dy + dz * dz) + 16.0f;
}

void padded_triangle_bounds(const float* p, float reach, float* bmin, float* bmax) {
    // Moller-Trumbore in f32 computes u, v with an absolute error of about
    // 6*eps*|tvec|*|e0|*|e1| / |det| (eps = 2^-24, tvec = origin - v0), and the
    // reference accepts |det| down to 1e-4 (triangle/mod.rs:140), so a ray can be
    // accepted while passing up to ~3.6e-3 * |tvec| * |e0|*|e1| outside the
    // triangle (for direction vectors of about unit length).  Capped at a quarter
    // of the longest edge: for large triangles the bound is only approached at
    // incidence angles within ~1e-5 rad of grazing.
    float e0[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]};
    float e1[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
    float e2[3] = {p[6] - p[3], p[7] - p[4], p[8] - p[5]};
    float l0 = std::sqrt(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2]);
    float l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    float l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    float lmax = std::max(l0, std::max(l1, l2));
    float fuzz = std::min(3.6e-3f * reach * l0 * l1, 0.25f * lmax);
    for (int a = 0; a < 3; a++) {
        float lo = std::min(p[a], std::min(p[3 + a], p[6 + a]));
        float hi = std::max(p[a], std::max(p[3 + a], p[6 + a]));
        float pad = fuzz + 1e-5f * std::max(std::fabs(lo), std::fabs(hi)) + 1e-6f;
        bmin[a] = lo - pad;
        bmax[a] = hi + pad;
    }
}

BuiltBvh build_bvh(const float* positions, uint64_t triangle_count, uint32_t max_leaf) {
    BuiltBvh out;
    if (max_leaf < 1) max_leaf = 1;
    if (max_leaf > (uint32_t)ptk::kLeafMaxTris) max_leaf = ptk::kLeafMaxTris;
    if (triangle_count == 0) {
        out.root_ref = ptk::kBvhSentinel;
        return out;
    }
    if (triangle_count >= (1ull << 28)) throw Error(PT_ERR_UNSUPPORTED, "mesh has too many triangles for the leaf encoding");
    Box raw;
    for (uint64_t i = 0; i < triangle_count; i++)
        for (int v = 0; v < 3; v++) raw.grow(positions + 9 * i + 3 * v, positions + 9 * i + 3 * v);
    float reach = padding_reach(raw.mn, raw.mx);
    std::vector<Ref> refs(triangle_count);
    for (uint64_t i = 0; i < triangle_count; i++) {
        padded_triangle_bounds(positions + 9 * i, reach, refs[i].bmin, refs[i].bmax);
        for (int k = 0; k < 3; k++) refs[i].c[k] = 0.5f * (refs[i].bmin[k] + refs[i].bmax[k]);
        refs[i].idx = (uint32_t)i;
    }
    Builder B{refs, max_leaf};
    Box bounds;
    SubTree tree;
    if (triangle_count < 200000) {
        out.root_ref = B.build(tree, 0, triangle_count, 0, &bounds);
    } else {
        // Top of the tree on this thread down to ~64 subtrees, each built by its own
        // task; subtrees are then appended in left-to-right order.
        struct Job {
            size_t lo, hi;
            int depth;
        };
        struct TopNode {
            int32_t left, right; // >= 0: top node, < 0: ~job index
        };
        std::vector<Job> jobs;
        std::vector<TopNode> top;
        std::function<int32_t(size_t, size_t, int)> carve = [&](size_t lo, size_t hi, int depth) -> int32_t {
            if (depth >= 6 || hi - lo <= 4096) {
                jobs.push_back({lo, hi, depth});
                return ~(int32_t)(jobs.size() - 1);
            }
            size_t mid = B.split(lo, hi, depth);
            int32_t me = (int32_t)top.size();
            top.push_back({0, 0});
            int32_t l = carve(lo, mid, depth + 1);
            int32_t r = carve(mid, hi, depth + 1);
            top[me].left = l;
            top[me].right = r;
            return me;
        };
        int32_t top_root = carve(0, triangle_count, 0);
        std::vector<SubTree> subs(jobs.size());
        std::vector<int32_t> sub_root(jobs.size());
        std::vector<Box> sub_box(jobs.size());
        std::vector<std::future<void>> futs;
        for (size_t j = 0; j < jobs.size(); j++)
            futs.push_back(std::async(std::launch::async, [&, j] {
                sub_root[j] = B.build(subs[j], jobs[j].lo, jobs[j].hi, jobs[j].depth, &sub_box[j]);
            }));
        for (auto& f : futs) f.get();
        // emit: top nodes first (pre-order), then each subtree
        std::function<int32_t(int32_t, Box*)> emit = [&](int32_t ref, Box* box) -> int32_t {
            if (ref < 0) {
                size_t j = (size_t)(~ref);
                *box = sub_box[j];
                return append_subtree(tree, subs[j], sub_root[j]);
            }
            int32_t me = (int32_t)tree.nodes.size();
            tree.nodes.emplace_back();
            Box lb, rb;
            int32_t l = emit(top[ref].left, &lb);
            int32_t r = emit(top[ref].right, &rb);
            ptk::BvhNode& nd = tree.nodes[me];
            for (int k = 0; k < 3; k++) {
                nd.lmin[k] = lb.mn[k];
                nd.lmax[k] = lb.mx[k];
                nd.rmin[k] = rb.mn[k];
                nd.rmax[k] = rb.mx[k];
            }
            nd.left = l;
            nd.right = r;
            nd.pad[0] = nd.pad[1] = 0;
            box->grow(lb.mn, lb.mx);
            box->grow(rb.mn, rb.mx);
            return me;
        };
        out.root_ref = emit(top_root, &bounds);
    }
    out.nodes = std::move(tree.nodes);
    out.order = std::move(tree.order);
    out.max_depth = tree.max_depth;
    for (int k = 0; k < 3; k++) {
        out.bmin[k] = bounds.mn[k];
        out.bmax[k] = bounds.mx[k];
    }
    if (out.max_depth > (uint32_t)ptk::kBvhMaxDepth) throw Error(PT_ERR_UNSUPPORTED, "BVH came out deeper than the traversal stack");
    return out;
}

WideBvh widen_bvh(const BuiltBvh& bvh) {
    WideBvh out;
    out.root_ref = bvh.root_ref;
    if (bvh.root_ref < 0 || bvh.root_ref == ptk::kBvhSentinel) return out; // empty mesh or a single leaf
    struct Child {
        int32_t ref;
        float mn[3], mx[3];
    };
    auto half_area = [](const Child& c) {
        float dx = c.mx[0] - c.mn[0], dy = c.mx[1] - c.mn[1], dz = c.mx[2] - c.mn[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto children_of = [&](int32_t node, Child* two) {
        const ptk::BvhNode& nd = bvh.nodes[node];
        two[0].ref = nd.left;
        two[1].ref = nd.right;
        memcpy(two[0].mn, nd.lmin, 12);
        memcpy(two[0].mx, nd.lmax, 12);
        memcpy(two[1].mn, nd.rmin, 12);
        memcpy(two[1].mx, nd.rmax, 12);
    };
    // returns {wide index, stack need below}
    std::function<uint32_t(int32_t, int32_t*)> emit = [&](int32_t node, int32_t* index) -> uint32_t {
        Child c[4];
        int n = 2;
        children_of(node, c);
        while (n < 4) {
            int pick = -1;
            float best = -1.0f;
            for (int i = 0; i < n; i++)
                if (c[i].ref >= 0 && half_area(c[i]) > best) {
                    best = half_area(c[i]);
                    pick = i;
                }
            if (pick < 0) break;
            Child two[2];
            children_of(c[pick].ref, two);
            c[pick] = two[0];
            c[n++] = two[1];
        }
        int32_t me = (int32_t)out.nodes.size();
        *index = me;
        out.nodes.emplace_back();
        ptk::WideNode w;
        memset(&w, 0, sizeof(w));
        double lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            lo[a] = c[0].mn[a];
            hi[a] = c[0].mx[a];
            for (int i = 1; i < n; i++) {
                lo[a] = std::min(lo[a], (double)c[i].mn[a]);
                hi[a] = std::max(hi[a], (double)c[i].mx[a]);
            }
        }
        for (int a = 0; a < 3; a++) {
            w.origin[a] = (float)lo[a]; // the smallest child minimum: a float already
            double extent = hi[a] - lo[a];
            int e = -126;
            if (extent > 0.0) e = std::max(-126, (int)std::ceil(std::log2(extent / 255.0)) - 1);
            for (;; e++) { // smallest step whose 255 cells cover the extent
                double step = std::ldexp(1.0, e);
                if (255.0 * step >= extent) break;
            }
            if (e > 127) throw Error(PT_ERR_UNSUPPORTED, "mesh extent too large for the node encoding");
            double step = std::ldexp(1.0, e);
            (a == 0 ? w.step_x : w.step_yz[a - 1]) = (float)step;
            for (int i = 0; i < n; i++) {
                double ql = std::floor(((double)c[i].mn[a] - lo[a]) / step);
                double qh = std::ceil(((double)c[i].mx[a] - lo[a]) / step);
                if (ql < 0.0) ql = 0.0;
                if (qh > 255.0 || ql > 255.0) throw Error(PT_ERR_UNSUPPORTED, "node quantisation out of range");
                // the decoded interval must contain the child's
                if (lo[a] + ql * step > (double)c[i].mn[a] || lo[a] + qh * step < (double)c[i].mx[a])
                    throw Error(PT_ERR_UNSUPPORTED, "node quantisation is not conservative");
                w.qlo[a] |= (uint32_t)ql << (8 * i);
                w.qhi[a] |= (uint32_t)qh << (8 * i);
            }
        }
        uint32_t need_below = 0;
        for (int i = 0; i < 4; i++) {
            if (i >= n) {
                w.child[i] = ptk::kBvhSentinel;
                continue;
            }
            if (c[i].ref >= 0) {
                int32_t idx;
                need_below = std::max(need_below, emit(c[i].ref, &idx));
                if (idx <= me) throw Error(PT_ERR_UNSUPPORTED, "wide node order"); // pre-order: cannot happen
                w.child[i] = idx;
            } else {
                w.child[i] = c[i].ref;
            }
        }
        out.nodes[me] = w;
        return (uint32_t)(n - 1) + need_below;
    };
    int32_t root;
    out.max_stack = emit(bvh.root_ref, &root);
    out.root_ref = root;
    return out;
}

std::vector<uint8_t> serialize_tree(const BuiltBvh& bvh, const float* positions, uint64_t triangle_count) {
    TreeHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "PTAMDTRE", 8);
    h.version = kTreeVersion;
    h.node_size = sizeof(ptk::BvhNode);
    h.triangle_count = triangle_count;
    h.node_count = bvh.nodes.size();
    h.root_ref = bvh.root_ref;
    h.max_depth = bvh.max_depth;
    memcpy(h.bmin, bvh.bmin, sizeof(h.bmin));
    memcpy(h.bmax, bvh.bmax, sizeof(h.bmax));
    h.geometry_hash = fnv1a(positions, (size_t)triangle_count * 9 * sizeof(float));
    std::vector<uint8_t> out(sizeof(h) + bvh.nodes.size() * sizeof(ptk::BvhNode) + bvh.order.size() * sizeof(uint32_t));
    uint8_t* p = out.data();
    memcpy(p, &h, sizeof(h));
    p += sizeof(h);
    if (!bvh.nodes.empty()) memcpy(p, bvh.nodes.data(), bvh.nodes.size() * sizeof(ptk::BvhNode));
    p += bvh.nodes.size() * sizeof(ptk::BvhNode);
    if (!bvh.order.empty()) memcpy(p, bvh.order.data(), bvh.order.size() * sizeof(uint32_t));
    return out;
}

bool deserialize_tree(const uint8_t* data, uint64_t size, const float* positions, uint64_t triangle_count,
                      BuiltBvh* out) {
    TreeHeader h;
    if (!data || size < sizeof(h)) return false;
    memcpy(&h, data, sizeof(h));
    if (memcmp(h.magic, "PTAMDTRE", 8) != 0 || h.version != kTreeVersion || h.node_size != sizeof(ptk::BvhNode))
        return false;
    if (h.triangle_count != triangle_count || triangle_count == 0 || triangle_count >= (1ull << 28)) return false;
    if (h.node_count > triangle_count) return false;
    uint64_t need = sizeof(h) + h.node_count * sizeof(ptk::BvhNode) + triangle_count * sizeof(uint32_t);
    if (size != need) return false;
    if (h.geometry_hash != fnv1a(positions, (size_t)triangle_count * 9 * sizeof(float))) return false;
    if (h.max_depth > (uint32_t)ptk::kBvhMaxDepth) return false;
    BuiltBvh b;
    b.nodes.resize(h.node_count);
    b.order.resize(triangle_count);
    const uint8_t* p = data + sizeof(h);
    if (h.node_count) memcpy(b.nodes.data(), p, h.node_count * sizeof(ptk::BvhNode));
    p += h.node_count * sizeof(ptk::BvhNode);
    memcpy(b.order.data(), p, triangle_count * sizeof(uint32_t));
    b.root_ref = h.root_ref;
    b.max_depth = h.max_depth;
    memcpy(b.bmin, h.bmin, sizeof(b.bmin));
    memcpy(b.bmax, h.bmax, sizeof(b.bmax));

    // Structural and conservativeness check: every triangle in exactly one leaf,
    // every node visited once, depth as promised, every box containing the padded
    // bounds of everything below it.
    Box raw;
    for (uint64_t i = 0; i < triangle_count; i++)
        for (int v = 0; v < 3; v++) raw.grow(positions + 9 * i + 3 * v, positions + 9 * i + 3 * v);
    float reach = padding_reach(raw.mn, raw.mx);
    std::vector<uint8_t> tri_seen(triangle_count, 0), node_seen(h.node_count, 0);
    struct Item {
        int32_t ref;
        uint32_t depth;
        float mn[3], mx[3];
    };
    std::vector<Item> stack;
    Item root;
    root.ref = b.root_ref;
    root.depth = 0;
    memcpy(root.mn, b.bmin, sizeof(root.mn));
    memcpy(root.mx, b.bmax, sizeof(root.mx));
    stack.push_back(root);
    uint64_t tris_covered = 0;
    auto inside = [](const float* mn, const float* mx, const float* omn, const float* omx) {
        for (int k = 0; k < 3; k++)
            if (!(mn[k] >= omn[k] && mx[k] <= omx[k])) return false;
        return true;
    };
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.depth > h.max_depth) return false;
        if (it.ref == ptk::kBvhSentinel) return false;
        if (it.ref < 0) {
            uint32_t raw_ref = (uint32_t)(~it.ref);
            uint64_t first = raw_ref >> 3, count = (raw_ref & 7u) + 1u;
            if (first + count > triangle_count) return false;
            for (uint64_t k = first; k < first + count; k++) {
                uint32_t idx = b.order[k];
                if (idx >= triangle_count || tri_seen[idx]) return false;
                tri_seen[idx] = 1;
                float mn[3], mx[3];
                padded_triangle_bounds(positions + 9 * (uint64_t)idx, reach, mn, mx);
                if (!inside(mn, mx, it.mn, it.mx)) return false;
            }
            tris_covered += count;
            continue;
        }
        if ((uint64_t)it.ref >= h.node_count || node_seen[it.ref]) return false;
        node_seen[it.ref] = 1;
        const ptk::BvhNode& nd = b.nodes[it.ref];
        if (!inside(nd.lmin, nd.lmax, it.mn, it.mx) || !inside(nd.rmin, nd.rmax, it.mn, it.mx)) return false;
        Item l, r;
        l.ref = nd.left;
        r.ref = nd.right;
        l.depth = r.depth = it.depth + 1;
        memcpy(l.mn, nd.lmin, sizeof(l.mn));
        memcpy(l.mx, nd.lmax, sizeof(l.mx));
        memcpy(r.mn, nd.rmin, sizeof(r.mn));
        memcpy(r.mx, nd.rmax, sizeof(r.mx));
        stack.push_back(l);
        stack.push_back(r);
    }
    if (tris_covered != triangle_count) return false;
    *out = std::move(b);
    return true;
}

} // namespace pt
