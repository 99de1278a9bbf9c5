// Host-side measure of the quality of a mesh's acceleration structure: builds the tree the library would build
// (host/bvh.cpp: build_bvh + widen_bvh), sends rays of the kinds a render sends through a CPU restatement of the
// device's search (pt_device.h: trav_inner / trav_leaf -- same child order, same pruning, same quantised boxes) and
// reports node records and triangle records fetched per search, next to the tree's surface-area cost.
//   g++ -O2 -std=c++17 -I pathtracing_amd/csrc tools/bvh_quality.cpp pathtracing_amd/csrc/host/bvh.cpp \
//       pathtracing_amd/csrc/host/obj_loader.cpp -pthread -o /tmp/bvh_quality
//   /tmp/bvh_quality /tmp/pathtracing_amd_dragon/dragon.obj [rays]
// Rays: (a) camera rays of the benchmark's view (origin (0,2,70) in mesh space, aimed at the mesh's box), (b) bounce
// rays leaving random triangles into the hemisphere of their normal, (c) rays from random points of a box around the
// mesh (the room's walls) into random inward directions.  Only rays that reach the mesh's bounds count as searches.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

} // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: bvh_quality MESH.obj [rays per kind]\n");
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
    pt::BuiltBvh bvh = pt::build_bvh(pos, nt, 4);
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
    return 0;
}
