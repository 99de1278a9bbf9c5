#include "front_end.hpp"

#include <algorithm>
#include <cstring>
#include <map>

namespace pt {

namespace {

struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t length = 0;
    uint8_t block[64];
    size_t fill = 0;

    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

    void compress(const uint8_t* p) {
        // floor(2^32 * |sin(i + 1)|) and the per-round shifts of RFC 1321
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
            0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
            0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
            0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
            0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
            0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
            0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
            0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                                  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t m[16];
        for (int i = 0; i < 16; i++)
            m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; i++) {
            uint32_t f;
            int g;
            if (i < 16) {
                f = (B & C) | (~B & D);
                g = i;
            } else if (i < 32) {
                f = (D & B) | (~D & C);
                g = (5 * i + 1) & 15;
            } else if (i < 48) {
                f = B ^ C ^ D;
                g = (3 * i + 5) & 15;
            } else {
                f = C ^ (B | ~D);
                g = (7 * i) & 15;
            }
            uint32_t t = D;
            D = C;
            C = B;
            B = B + rotl(A + f + K[i] + m[g], S[i]);
            A = t;
        }
        a += A;
        b += B;
        c += C;
        d += D;
    }

    void update(const uint8_t* data, size_t size) {
        length += size;
        while (size) {
            size_t take = std::min(size, sizeof(block) - fill);
            memcpy(block + fill, data, take);
            fill += take;
            data += take;
            size -= take;
            if (fill == 64) {
                compress(block);
                fill = 0;
            }
        }
    }

    std::string finish() {
        uint64_t bits = length * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (8 * i));
        update(len, 8);
        static const char* hex = "0123456789abcdef";
        std::string out;
        for (uint32_t w : {a, b, c, d})
            for (int i = 0; i < 4; i++) {
                uint8_t byte = (uint8_t)(w >> (8 * i));
                out.push_back(hex[byte >> 4]);
                out.push_back(hex[byte & 15]);
            }
        return out;
    }
};

} // namespace

std::string md5_hex(const uint8_t* data, size_t size) {
    Md5 h;
    if (size) h.update(data, size);
    return h.finish();
}

std::string scene_md5(const FileStore& store, const std::string& scene_path) {
    std::unique_ptr<HostScene> scene = load_scene(store, scene_path);
    std::map<std::string, std::string> digests; // sorted by path like the client's sort_by (byte order of the strings)
    auto add = [&](const std::string& path) {
        if (digests.count(path)) return;
        std::vector<uint8_t> data = store.fetch_file(path);
        digests[path] = md5_hex(data.data(), data.size());
    };
    add(scene_path);
    for (const std::string& p : scene->mesh_paths) add(p);
    for (const std::string& p : scene->material_paths) add(p);
    for (const std::string& p : scene->image_paths) add(p);
    std::string acc;
    for (const auto& kv : digests) {
        std::string joined = acc + kv.second;
        acc = md5_hex(reinterpret_cast<const uint8_t*>(joined.data()), joined.size());
    }
    return acc;
}

std::vector<uint8_t> discovery_response_encode(uint16_t websocket_port) {
    std::vector<uint8_t> out;
    uint32_t v = websocket_port;
    while (v >= 0x80u) {
        out.push_back((uint8_t)(v | 0x80u));
        v >>= 7;
    }
    out.push_back((uint8_t)v);
    return out;
}

bool discovery_response_decode(const uint8_t* bytes, size_t len, uint16_t* websocket_port) {
    uint32_t v = 0;
    for (size_t i = 0; i < 3; i++) { // a u16 varint has at most three bytes
        if (i >= len) return false;  // ran out of input inside the number
        const uint8_t byte = bytes[i];
        if (i == 2 && byte > 0x03) return false; // more than sixteen bits, or a fourth byte announced
        v |= (uint32_t)(byte & 0x7Fu) << (7 * i);
        if (!(byte & 0x80u)) {
            *websocket_port = (uint16_t)v;
            return true;
        }
    }
    return false;
}

} // namespace pt
