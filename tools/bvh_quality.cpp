// Host-side measure of the quality of a mesh's acceleration structure: builds the tree the library would build
// (host/bvh.cpp: build_bvh + widen_bvh), sends rays of the kinds a render sends through a CPU restatement of the
// device's search (pt_device.h: trav_inner / trav_leaf -- same child order, same pruning, same quantised boxes) and
// reports node records and triangle records fetched per search, next to the tree's surface-area cost.
//   g++ -O2 -std=c++17 -I pathtracing_amd/csrc -I include tools/bvh_quality.cpp pathtracing_amd/csrc/host/bvh.cpp,
//       then pathtracing_amd/csrc/host/obj_loader.cpp -pthread -o /tmp/bvh_quality (one command line)
//   /tmp/bvh_quality /tmp/pathtracing_amd_dragon/dragon.obj [rays per kind] [most triangles per leaf]
// Rays: (a) camera rays of the benchmark's view (origin (0,2,70) in mesh space, aimed at the mesh's box), (b) bounce
// rays leaving random triangles into the hemisphere of their normal, (c) rays from random points of a box around the
// mesh (the room's walls) into random inward directions.  Only rays that reach the mesh's bounds count as searches.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "host/bvh.hpp"

namespace pt {
void set_last_error(const std::string&) {}
} // namespace pt

namespace {

struct Ray {
    float o[3], d[3];
};

struct Stats {
    double nodes = 0, tris = 0, leaves = 0, searches = 0, hits = 0, max_stack = 0;
};

bool triangle_hit(const float* p, const Ray& r, float* t_out) {
    float e0[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e1[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
    float pv[3] = {r.d[1] * e1[2] - r.d[2] * e1[1], r.d[2] * e1[0] - r.d[0] * e1[2], r.d[0] * e1[1] - r.d[1] * e1[0]};
    float det = e0[0] * pv[0] + e0[1] * pv[1] + e0[2] * pv[2];
    if (det < 1e-4f && det > -1e-4f) return false;
    float inv = 1.0f / det;
    float tv[3] = {r.o[0] - p[0], r.o[1] - p[1], r.o[2] - p[2]};
    float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    float q[3] = {tv[1] * e0[2] - tv[2] * e0[1], tv[2] * e0[0] - tv[0] * e0[2], tv[0] * e0[1] - tv[1] * e0[0]};
    float v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = (e1[0] * q[0] + e1[1] * q[1] + e1[2] * q[2]) * inv;
    if (t < 1e-4f) return false;
    *t_out = t;
    return true;
}

// The device's search (trav_inner + trav_leaf) on the host, counting what it fetches.
void search(const pt::WideBvh& w, const std::vector<uint32_t>& order, const float* positions, const Ray& r, Stats& s) {
    float inv[3];
    for (int k = 0; k < 3; k++) inv[k] = std::min(1e28f, std::max(-1e28f, 1.0f / r.d[k]));
    float tbest = INFINITY;
    int32_t stack[128];
    int sp = 0;
    int32_t cur = w.root_ref;
    double max_sp = 0;
    while (cur != ptk::kBvhSentinel) {
        if (cur >= 0) {
            const ptk::WideNode& n = w.nodes[(size_t)cur];
            s.nodes++;
            float step[3] = {n.step_x, n.step_yz[0], n.step_yz[1]};
            float t[4];
            int32_t ref[4];
            for (int c = 0; c < 4; c++) {
                ref[c] = n.child[c];
                float lo = 0.0f, hi = tbest;
                for (int k = 0; k < 3; k++) {
                    float a = step[k] * inv[k], b = (n.origin[k] - r.o[k]) * inv[k];
                    float slack = 1.1920929e-7f * (512.0f * std::fabs(a) + 4.0f * std::fabs(b));
                    float ql = (float)((n.qlo[k] >> (8 * c)) & 255u), qh = (float)((n.qhi[k] >> (8 * c)) & 255u);
                    float nq = inv[k] < 0.0f ? qh : ql, fq = inv[k] < 0.0f ? ql : qh;
                    float tn = nq * a + (b - slack), tf = fq * a + (b + slack);
                    if (tn == tn) lo = std::max(lo, tn);
                    if (tf == tf) hi = std::min(hi, tf);
                }
                t[c] = (lo <= hi && ref[c] != ptk::kBvhSentinel) ? lo : INFINITY;
            }
            auto order_pair = [&](int a, int b) {
                if (t[b] < t[a]) {
                    std::swap(t[a], t[b]);
                    std::swap(ref[a], ref[b]);
                }
            };
            order_pair(0, 1);
            order_pair(2, 3);
            order_pair(0, 2);
            order_pair(1, 3);
            order_pair(1, 2);
            if (t[0] == INFINITY) {
                cur = sp ? stack[--sp] : ptk::kBvhSentinel;
                continue;
            }
            if (t[3] != INFINITY) stack[sp++] = ref[3];
            if (t[2] != INFINITY) stack[sp++] = ref[2];
            if (t[1] != INFINITY) stack[sp++] = ref[1];
            max_sp = std::max(max_sp, (double)sp);
            cur = ref[0];
        } else {
            uint32_t raw = (uint32_t)(~cur), first = raw >> 3, count = (raw & 7u) + 1u;
            s.leaves++;
            for (uint32_t k = 0; k < count; k++) {
                s.tris++;
                float t;
                if (triangle_hit(positions + 9 * (size_t)order[first + k], r, &t) && t < tbest) tbest = t;
            }
            cur = sp ? stack[--sp] : ptk::kBvhSentinel;
        }
    }
    s.searches++;
    s.hits += tbest < INFINITY ? 1 : 0;
    s.max_stack = std::max(s.max_stack, max_sp);
}

bool reaches(const float* bmin, const float* bmax, const Ray& r) {
    float lo = 0.0f, hi = INFINITY;
    for (int k = 0; k < 3; k++) {
        float inv = 1.0f / r.d[k];
        float a = (bmin[k] - r.o[k]) * inv, b = (bmax[k] - r.o[k]) * inv;
        lo = std::max(lo, std::min(a, b));
        hi = std::min(hi, std::max(a, b));
    }
    return lo <= hi;
}

// Surface-area cost of the wide tree as the device searches it: expected node records and triangle records
// fetched by a random line through the root's box (no occlusion).
void sah_cost(const pt::WideBvh& w, double* node_cost, double* tri_cost) {
    struct Item {
        int32_t node;
        double area;
    };
    auto half_area = [](const double* lo, const double* hi) {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    if (w.root_ref < 0 || w.nodes.empty()) {
        *node_cost = *tri_cost = 0;
        return;
    }
    double root_area = 0;
    {
        const ptk::WideNode& n = w.nodes[0];
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        float step[3] = {n.step_x, n.step_yz[0], n.step_yz[1]};
        for (int c = 0; c < 4; c++) {
            if (n.child[c] == ptk::kBvhSentinel) continue;
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(lo[k], n.origin[k] + (double)((n.qlo[k] >> (8 * c)) & 255u) * step[k]);
                hi[k] = std::max(hi[k], n.origin[k] + (double)((n.qhi[k] >> (8 * c)) & 255u) * step[k]);
            }
        }
        root_area = half_area(lo, hi);
    }
    double nodes = 1.0, tris = 0.0;
    std::vector<Item> st{{0, root_area}};
    while (!st.empty()) {
        Item it = st.back();
        st.pop_back();
        const ptk::WideNode& n = w.nodes[(size_t)it.node];
        float step[3] = {n.step_x, n.step_yz[0], n.step_yz[1]};
        for (int c = 0; c < 4; c++) {
            if (n.child[c] == ptk::kBvhSentinel) continue;
            double lo[3], hi[3];
            for (int k = 0; k < 3; k++) {
                lo[k] = n.origin[k] + (double)((n.qlo[k] >> (8 * c)) & 255u) * step[k];
                hi[k] = n.origin[k] + (double)((n.qhi[k] >> (8 * c)) & 255u) * step[k];
            }
            double p = half_area(lo, hi) / root_area;
            if (n.child[c] >= 0) {
                nodes += p;
                st.push_back({n.child[c], 0});
            } else {
                tris += p * (double)((((uint32_t)~n.child[c]) & 7u) + 1u);
            }
        }
    }
    *node_cost = nodes;
    *tri_cost = tris;
}


// ---- what wider nodes would visit: the same binary tree cut into nodes of up to W children by the same
// minimum-surface-area-cost rule (host/bvh.cpp: WidePlan, here for any W), searched nearest child first with
// exact boxes.  W = 4 through this route shows what the device's 8-bit boxes cost; W = 8 is the estimate for
// a 128-byte node.
struct NaryNode {
    int n = 0;
    float mn[8][3], mx[8][3];
    int32_t child[8]; // >= 0: NaryNode index, < 0: leaf reference as in the binary tree
};

std::vector<NaryNode> cut_wide(const pt::BuiltBvh& bvh, int W, float tri_cost) {
    const size_t n = bvh.nodes.size();
    const int S = W - 1; // slots a subtree can be given by the node above
    std::vector<float> cost(n * S), area(n);
    std::vector<uint8_t> choice(n * S);
    auto ha = [](const float* mn, const float* mx) {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto side = [&](int32_t ref, const float* mn, const float* mx, float* c) {
        if (ref < 0) {
            float v = ha(mn, mx) * (float)((((uint32_t)~ref) & 7u) + 1u) * tri_cost;
            for (int i = 0; i < S; i++) c[i] = v;
        } else {
            for (int i = 0; i < S; i++) c[i] = cost[(size_t)ref * S + i];
        }
    };
    for (size_t k = n; k-- > 0;) {
        const ptk::BvhNode& nd = bvh.nodes[k];
        float cl[8], cr[8], mn[3], mx[3];
        side(nd.left, nd.lmin, nd.lmax, cl);
        side(nd.right, nd.rmin, nd.rmax, cr);
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(nd.lmin[a], nd.rmin[a]);
            mx[a] = std::max(nd.lmax[a], nd.rmax[a]);
        }
        area[k] = ha(mn, mx);
        float spread[9];
        uint8_t at[9];
        for (int slots = 2; slots <= W; slots++) {
            spread[slots] = INFINITY;
            at[slots] = 1;
            for (int j = 1; j < slots; j++) {
                if (j > S || slots - j > S) continue;
                float c = cl[j - 1] + cr[slots - j - 1];
                if (c < spread[slots]) {
                    spread[slots] = c;
                    at[slots] = (uint8_t)j;
                }
            }
        }
        cost[k * S] = area[k] + spread[W];
        choice[k * S] = at[W];
        for (int i = 2; i <= S; i++) {
            if (spread[i] < cost[k * S + i - 2]) {
                cost[k * S + i - 1] = spread[i];
                choice[k * S + i - 1] = at[i];
            } else {
                cost[k * S + i - 1] = cost[k * S + i - 2];
                choice[k * S + i - 1] = 0;
            }
        }
    }
    std::vector<NaryNode> out;
    struct Child {
        int32_t ref;
        const float *mn, *mx;
    };
    std::function<void(Child, int, NaryNode&)> spread_into = [&](Child c, int slots, NaryNode& w) {
        int j = c.ref < 0 ? 0 : choice[(size_t)c.ref * S + slots - 1];
        if (c.ref < 0 || slots == 1) {
            memcpy(w.mn[w.n], c.mn, 12);
            memcpy(w.mx[w.n], c.mx, 12);
            w.child[w.n++] = c.ref; // inner references are resolved by the caller
            return;
        }
        if (j == 0) return spread_into(c, slots - 1, w);
        const ptk::BvhNode& nd = bvh.nodes[(size_t)c.ref];
        spread_into({nd.left, nd.lmin, nd.lmax}, j, w);
        spread_into({nd.right, nd.rmin, nd.rmax}, slots - j, w);
    };
    std::function<int32_t(int32_t)> emit = [&](int32_t node) -> int32_t {
        int32_t me = (int32_t)out.size();
        out.emplace_back();
        NaryNode w;
        const ptk::BvhNode& nd = bvh.nodes[(size_t)node];
        int j = choice[(size_t)node * S];
        spread_into({nd.left, nd.lmin, nd.lmax}, j, w);
        spread_into({nd.right, nd.rmin, nd.rmax}, W - j, w);
        for (int i = 0; i < w.n; i++)
            if (w.child[i] >= 0) w.child[i] = emit(w.child[i]);
        out[(size_t)me] = w;
        return me;
    };
    if (bvh.root_ref >= 0 && bvh.root_ref != ptk::kBvhSentinel) emit(bvh.root_ref);
    return out;
}

void search_wide(const std::vector<NaryNode>& nodes, const std::vector<uint32_t>& order, const float* positions, const Ray& r,
                 Stats& s) {
    float inv[3];
    for (int k = 0; k < 3; k++) inv[k] = std::min(1e28f, std::max(-1e28f, 1.0f / r.d[k]));
    float tbest = INFINITY;
    int32_t stack[256];
    int sp = 0;
    int32_t cur = 0;
    while (cur != ptk::kBvhSentinel) {
        if (cur >= 0) {
            const NaryNode& n = nodes[(size_t)cur];
            s.nodes++;
            float t[8];
            int idx[8], hits = 0;
            for (int c = 0; c < n.n; c++) {
                float lo = 0.0f, hi = tbest;
                for (int k = 0; k < 3; k++) {
                    float a = (n.mn[c][k] - r.o[k]) * inv[k], b = (n.mx[c][k] - r.o[k]) * inv[k];
                    lo = std::max(lo, std::min(a, b));
                    hi = std::min(hi, std::max(a, b));
                }
                if (lo <= hi * 1.0000004f) {
                    t[hits] = lo;
                    idx[hits++] = c;
                }
            }
            for (int i = 1; i < hits; i++) // farthest first, so the nearest is popped first
                for (int j = i; j > 0 && t[j] > t[j - 1]; j--) {
                    std::swap(t[j], t[j - 1]);
                    std::swap(idx[j], idx[j - 1]);
                }
            for (int i = 0; i < hits; i++) stack[sp++] = n.child[idx[i]];
            cur = sp ? stack[--sp] : ptk::kBvhSentinel;
        } else {
            uint32_t raw = (uint32_t)(~cur), first = raw >> 3, count = (raw & 7u) + 1u;
            s.leaves++;
            for (uint32_t k = 0; k < count; k++) {
                s.tris++;
                float t;
                if (triangle_hit(positions + 9 * (size_t)order[first + k], r, &t) && t < tbest) tbest = t;
            }
            cur = sp ? stack[--sp] : ptk::kBvhSentinel;
        }
    }
    s.searches++;
}

} // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: bvh_quality MESH.obj [rays per kind] [most triangles per leaf]\n");
        return 2;
    }
    const size_t per_kind = argc > 2 ? (size_t)atol(argv[2]) : 200000;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> text;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.insert(text.end(), buf, buf + n);
    fclose(f);
    pt::HostMesh mesh = pt::load_obj(text.data(), text.size());
    const float* pos = mesh.positions.data();
    const uint64_t nt = mesh.triangle_count;
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t max_leaf = argc > 3 ? (uint32_t)atoi(argv[3]) : 4;
    pt::BuiltBvh bvh = pt::build_bvh(pos, nt, max_leaf);
    auto t1 = std::chrono::steady_clock::now();
    pt::WideBvh wide = pt::widen_bvh(bvh);
    auto t2 = std::chrono::steady_clock::now();
    double node_cost, tri_cost;
    sah_cost(wide, &node_cost, &tri_cost);
    size_t leaves = 0, slots = 0;
    for (const ptk::WideNode& wn : wide.nodes)
        for (int c = 0; c < 4; c++) {
            slots += wn.child[c] != ptk::kBvhSentinel;
            leaves += wn.child[c] < 0;
        }
    printf("triangles %llu  binary nodes %zu  wide nodes %zu (%.2f children each)  leaves %zu (%.2f triangles each)  depth %u  stack %u\n",
           (unsigned long long)nt, bvh.nodes.size(), wide.nodes.size(), (double)slots / wide.nodes.size(), leaves,
           (double)nt / leaves, bvh.max_depth, wide.max_stack);
    printf("build %.2f s  widen %.2f s   surface-area cost: %.2f node records + %.2f triangle records per random line\n",
           std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), node_cost, tri_cost);

    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    auto unit = [&](float* d) {
        float z = 2.0f * U(rng) - 1.0f, phi = 6.2831853f * U(rng), s = std::sqrt(std::max(0.0f, 1.0f - z * z));
        d[0] = s * std::cos(phi);
        d[1] = s * std::sin(phi);
        d[2] = z;
    };
    const char* names[3] = {"camera rays", "bounce rays off the mesh", "rays from the room's walls"};
    Stats total;
    std::vector<Ray> all_rays;
    for (int kind = 0; kind < 3; kind++) {
        Stats s;
        size_t made = 0, tried = 0;
        while (made < per_kind && tried < per_kind * 50) {
            tried++;
            Ray r;
            if (kind == 0) {
                // the benchmark's camera: (0, 0, 70) looking down -z, the mesh moved by (0, -2, 0): in mesh space the eye is at (0, 2, 70)
                r.o[0] = 0.0f; r.o[1] = 2.0f; r.o[2] = 70.0f;
                float half_w = 70.0f * std::tan(0.523f / 2.0f), half_h = half_w * 1080.0f / 1920.0f;
                float tx = (2.0f * U(rng) - 1.0f) * half_w, ty = (2.0f * U(rng) - 1.0f) * half_h + 2.0f;
                float d[3] = {tx - r.o[0], ty - r.o[1], -70.0f};
                float l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                for (int k = 0; k < 3; k++) r.d[k] = d[k] / l;
            } else if (kind == 1) {
                size_t tri = (size_t)(U(rng) * (float)nt) % nt;
                const float* p = pos + 9 * tri;
                float a = U(rng), b = U(rng);
                if (a + b > 1.0f) { a = 1.0f - a; b = 1.0f - b; }
                float e0[3], e1[3], nrm[3];
                for (int k = 0; k < 3; k++) { e0[k] = p[3 + k] - p[k]; e1[k] = p[6 + k] - p[k]; }
                nrm[0] = e0[1] * e1[2] - e0[2] * e1[1]; nrm[1] = e0[2] * e1[0] - e0[0] * e1[2]; nrm[2] = e0[0] * e1[1] - e0[1] * e1[0];
                unit(r.d);
                float side = r.d[0] * nrm[0] + r.d[1] * nrm[1] + r.d[2] * nrm[2];
                // the mesh's triangles face outwards or inwards consistently; take both hemispheres in turn like a two-sided surface
                if ((side < 0.0f) != ((tried & 1u) != 0u)) for (int k = 0; k < 3; k++) r.d[k] = -r.d[k];
                for (int k = 0; k < 3; k++) r.o[k] = p[k] + a * e0[k] + b * e1[k] + 1e-4f * r.d[k];
            } else {
                // a point of the room's walls (half-width 16, centred on the mesh's transform) and a direction into the room
                const float half = 16.0f;
                int face = (int)(U(rng) * 6.0f) % 6, axis = face >> 1;
                float sgn = (face & 1) ? 1.0f : -1.0f;
                for (int k = 0; k < 3; k++) r.o[k] = (2.0f * U(rng) - 1.0f) * half;
                r.o[axis] = sgn * half;
                r.o[1] += 2.0f;
                unit(r.d);
                if (r.d[axis] * sgn > 0.0f) r.d[axis] = -r.d[axis];
            }
            if (!reaches(bvh.bmin, bvh.bmax, r)) continue;
            made++;
            all_rays.push_back(r);
            search(wide, bvh.order, pos, r, s);
        }
        printf("%-28s %8.0f searches (%5.1f %% of the rays tried), %5.1f %% hit: %6.2f node records  %6.2f triangle records  %5.2f leaves  per search; deepest stack %.0f\n",
               names[kind], s.searches, 100.0 * s.searches / tried, 100.0 * s.hits / s.searches, s.nodes / s.searches,
               s.tris / s.searches, s.leaves / s.searches, s.max_stack);
        total.nodes += s.nodes;
        total.tris += s.tris;
        total.leaves += s.leaves;
        total.searches += s.searches;
    }
    printf("%-28s %8.0f searches: %6.2f node records  %6.2f triangle records  %5.2f leaves  per search  -> %.0f B per search (64 B nodes, 48 B triangles)\n",
           "all", total.searches, total.nodes / total.searches, total.tris / total.searches, total.leaves / total.searches,
           (64.0 * total.nodes + 48.0 * total.tris) / total.searches);
    const float cut_tri_cost = pt::default_bvh_tuning().tri_cost;
    for (int W : {4, 8}) {
        std::vector<NaryNode> nodes = cut_wide(bvh, W, cut_tri_cost);
        if (nodes.empty()) continue;
        Stats s;
        for (const Ray& r : all_rays) search_wide(nodes, bvh.order, pos, r, s);
        size_t kids = 0;
        for (const NaryNode& n : nodes) kids += (size_t)n.n;
        printf("%d-wide nodes with exact boxes: %zu nodes (%.2f children each): %6.2f node records  %6.2f triangle records  %5.2f leaves  per search\n",
               W, nodes.size(), (double)kids / nodes.size(), s.nodes / s.searches, s.tris / s.searches, s.leaves / s.searches);
    }
    return 0;
}
