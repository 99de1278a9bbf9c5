// Wavefront OBJ -> triangle list, with the behaviour the reference gets from
// tobj 3.2.5's load_obj_buf under
//   LoadOptions { triangulate: true, ..OFFLINE_RENDERING_LOAD_OPTIONS }
// (worker/src/scene/resource/mesh/obj_loader.rs:9-82):
//   * `v` keeps its first three numbers, `vt` its first two, `vn` three;
//   * position / texcoord / normal indices are separate streams (single_index
//     off); 1-based, negative = relative to the elements read so far;
//   * points and lines are dropped; a polygon a b c d ... becomes the fan
//     (a,b,c) (a,c,d) ...;
//   * `o` / `g` close the current model; models are concatenated in file order; materials are
//     ignored (the reference's loader callback always fails);
//   * inside one model the k-th group of three normal (texcoord) indices belongs
//     to the k-th triangle; a model with none has no normals (texcoords).
// tobj's source is not part of the reference checkout: the triangulation order
// is restated from its documented behaviour and is not pinned by any reference
// test.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"

namespace pt {
namespace {

constexpr int64_t kMissing = INT64_MIN;

struct FaceVertex {
    int64_t v = kMissing, vt = kMissing, vn = kMissing;
};

struct Model {
    std::vector<uint32_t> pos_idx, uv_idx, nrm_idx; // 3 per triangle (uv/nrm may be shorter)
};

struct Cursor {
    const char* p;
    const char* end;
    void skip_space() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) p++;
    }
    bool at_end() {
        skip_space();
        return p >= end;
    }
    // next whitespace-delimited word
    bool word(const char** b, const char** e) {
        skip_space();
        if (p >= end) return false;
        *b = p;
        while (p < end && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) p++;
        *e = p;
        return true;
    }
};

bool parse_float(const char* b, const char* e, float* out) {
    char buf[64];
    size_t n = (size_t)(e - b);
    if (n == 0 || n >= sizeof(buf)) return false;
    memcpy(buf, b, n);
    buf[n] = 0;
    char* endp = nullptr;
    float v = strtof(buf, &endp); // correctly rounded decimal -> f32, like str::parse::<f32>
    if (endp != buf + n) return false;
    *out = v;
    return true;
}

bool parse_int(const char* b, const char* e, int64_t* out) {
    if (b == e) return false;
    bool neg = false;
    const char* p = b;
    if (*p == '-' || *p == '+') {
        neg = *p == '-';
        p++;
    }
    if (p == e) return false;
    int64_t v = 0;
    for (; p < e; p++) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
        if (v > (int64_t)1 << 40) return false;
    }
    *out = neg ? -v : v;
    return true;
}

// "v", "v/vt", "v//vn", "v/vt/vn"
bool parse_face_vertex(const char* b, const char* e, FaceVertex* fv) {
    const char* parts[3] = {nullptr, nullptr, nullptr};
    const char* ends[3] = {nullptr, nullptr, nullptr};
    int n = 0;
    const char* s = b;
    for (const char* p = b; p <= e; p++) {
        if (p == e || *p == '/') {
            if (n >= 3) return false;
            parts[n] = s;
            ends[n] = p;
            n++;
            s = p + 1;
        }
    }
    if (n == 0 || parts[0] == ends[0]) return false;
    if (!parse_int(parts[0], ends[0], &fv->v)) return false;
    if (n > 1 && parts[1] != ends[1] && !parse_int(parts[1], ends[1], &fv->vt)) return false;
    if (n > 2 && parts[2] != ends[2] && !parse_int(parts[2], ends[2], &fv->vn)) return false;
    return true;
}

// OBJ index (1-based or negative-relative) -> 0-based, checked against `count`.
uint32_t resolve(int64_t idx, size_t count, const char* what) {
    int64_t r = idx > 0 ? idx - 1 : (int64_t)count + idx;
    if (idx == 0 || r < 0 || (size_t)r >= count)
        throw Error(PT_ERR_PARSE, std::string("OBJ: ") + what + " index out of range");
    return (uint32_t)r;
}

} // namespace

HostMesh load_obj(const uint8_t* data, size_t size) {
    std::vector<float> pos, uv, nrm; // 3, 2, 3 floats per element
    std::vector<Model> models;
    Model cur;
    bool cur_has_faces = false;

    auto close_model = [&]() {
        if (cur_has_faces) models.push_back(std::move(cur));
        cur = Model();
        cur_has_faces = false;
    };

    const char* p = reinterpret_cast<const char*>(data);
    const char* end = p + size;
    std::vector<FaceVertex> face;
    while (p < end) {
        const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        if (!eol) eol = end;
        Cursor c{p, eol};
        p = eol < end ? eol + 1 : end;
        const char *wb, *we;
        if (!c.word(&wb, &we)) continue;
        size_t wl = (size_t)(we - wb);
        if (wb[0] == '#') continue;
        if (wl == 1 && wb[0] == 'v') {
            float f[3];
            for (int k = 0; k < 3; k++) {
                const char *b, *e;
                if (!c.word(&b, &e) || !parse_float(b, e, &f[k]))
                    throw Error(PT_ERR_PARSE, "OBJ: malformed 'v' line");
            }
            pos.insert(pos.end(), f, f + 3);
        } else if (wl == 2 && wb[0] == 'v' && wb[1] == 't') {
            float f[2];
            for (int k = 0; k < 2; k++) {
                const char *b, *e;
                if (!c.word(&b, &e) || !parse_float(b, e, &f[k]))
                    throw Error(PT_ERR_PARSE, "OBJ: malformed 'vt' line");
            }
            uv.insert(uv.end(), f, f + 2);
        } else if (wl == 2 && wb[0] == 'v' && wb[1] == 'n') {
            float f[3];
            for (int k = 0; k < 3; k++) {
                const char *b, *e;
                if (!c.word(&b, &e) || !parse_float(b, e, &f[k]))
                    throw Error(PT_ERR_PARSE, "OBJ: malformed 'vn' line");
            }
            nrm.insert(nrm.end(), f, f + 3);
        } else if (wl == 1 && (wb[0] == 'f' || wb[0] == 'l')) {
            face.clear();
            const char *b, *e;
            while (c.word(&b, &e)) {
                FaceVertex fv;
                if (!parse_face_vertex(b, e, &fv)) throw Error(PT_ERR_PARSE, "OBJ: malformed face vertex");
                face.push_back(fv);
            }
            if (face.empty()) throw Error(PT_ERR_PARSE, "OBJ: face without vertices");
            if (face.size() < 3) continue; // points and lines are ignored
            auto emit = [&](const FaceVertex& fv) {
                cur.pos_idx.push_back(resolve(fv.v, pos.size() / 3, "position"));
                if (!uv.empty() && fv.vt != kMissing) cur.uv_idx.push_back(resolve(fv.vt, uv.size() / 2, "texcoord"));
                if (!nrm.empty() && fv.vn != kMissing) cur.nrm_idx.push_back(resolve(fv.vn, nrm.size() / 3, "normal"));
            };
            for (size_t k = 2; k < face.size(); k++) {
                emit(face[0]);
                emit(face[k - 1]);
                emit(face[k]);
            }
            cur_has_faces = true;
        } else if (wl == 1 && (wb[0] == 'o' || wb[0] == 'g')) {
            close_model();
        }
        // everything else does not affect the triangles.  `usemtl` would split a
        // model only when it names a different *loaded* material, and the reference's
        // material loader always fails, so it never splits; mtllib, s, comments: ignored.
    }
    close_model();

    HostMesh mesh;
    uint64_t total = 0;
    bool any_nrm = false, any_uv = false;
    for (const Model& m : models) {
        total += m.pos_idx.size() / 3;
        any_nrm |= !m.nrm_idx.empty();
        any_uv |= !m.uv_idx.empty();
    }
    mesh.triangle_count = total;
    mesh.positions.resize(total * 9);
    if (any_nrm) {
        mesh.normals.assign(total * 9, 0.0f);
        mesh.normal_valid.assign(total, 0);
    }
    if (any_uv) mesh.uvs.assign(total * 6, 0.0f);
    uint64_t t = 0;
    for (const Model& m : models) {
        size_t ntri = m.pos_idx.size() / 3;
        // chunks(3) of a stream whose length is not a multiple of three ends in a
        // short chunk that the reference's try_into().unwrap() rejects
        if (m.nrm_idx.size() % 3 != 0 || m.uv_idx.size() % 3 != 0)
            throw Error(PT_ERR_PARSE, "OBJ: model mixes vertices with and without normals/texcoords");
        for (size_t k = 0; k < ntri; k++, t++) {
            for (int j = 0; j < 3; j++) {
                uint32_t pi = m.pos_idx[3 * k + j];
                mesh.positions[t * 9 + 3 * j + 0] = pos[3 * pi + 0];
                mesh.positions[t * 9 + 3 * j + 1] = pos[3 * pi + 1];
                mesh.positions[t * 9 + 3 * j + 2] = pos[3 * pi + 2];
            }
            if (3 * k + 2 < m.nrm_idx.size()) {
                mesh.normal_valid[t] = 1;
                for (int j = 0; j < 3; j++) {
                    uint32_t ni = m.nrm_idx[3 * k + j];
                    mesh.normals[t * 9 + 3 * j + 0] = nrm[3 * ni + 0];
                    mesh.normals[t * 9 + 3 * j + 1] = nrm[3 * ni + 1];
                    mesh.normals[t * 9 + 3 * j + 2] = nrm[3 * ni + 2];
                }
            }
            if (3 * k + 2 < m.uv_idx.size()) {
                for (int j = 0; j < 3; j++) {
                    uint32_t ui = m.uv_idx[3 * k + j];
                    mesh.uvs[t * 6 + 2 * j + 0] = uv[2 * ui + 0];
                    mesh.uvs[t * 6 + 2 * j + 1] = uv[2 * ui + 1];
                }
            }
        }
    }
    return mesh;
}

} // namespace pt
