// PNG -> RGBA8, the conversion the reference gets from
// image::load_from_memory(..).to_rgba8() (worker/src/scene/resource/image.rs:13-19).
// Only the colour channels are ever read (image.rs:39-46), so alpha handling is
// kept simple: opaque unless the file carries an alpha channel or a tRNS palette.
// Supported: non-interlaced, bit depth 8 for colour types 0/2/4/6 and bit depth
// 1/2/4/8 for palette images (type 3) -- this covers the shipped textures
// (test_texture_0.png: 4096x4096 RGBA8; test_texture_1.png: 1024x1024 palette).
// gAMA / sRGB / iCCP chunks are ignored, as the `image` crate ignores them.
// The `image` / `png` crates are not part of the reference checkout; this decode
// follows the PNG specification and is not pinned by any reference test.
#include <zlib.h>

#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"

namespace pt {
namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = p > a ? p - a : a - p;
    int pb = p > b ? p - b : b - p;
    int pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

} // namespace

HostImage decode_png(const uint8_t* data, size_t size) {
    static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || memcmp(data, kSig, 8) != 0) throw Error(PT_ERR_PARSE, "image: not a PNG file");
    size_t off = 8;
    uint32_t width = 0, height = 0;
    int bit_depth = 0, color_type = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool have_ihdr = false, have_iend = false;
    while (off + 12 <= size && !have_iend) {
        uint32_t len = be32(data + off);
        const uint8_t* type = data + off + 4;
        if ((size_t)len > size - off - 12) throw Error(PT_ERR_PARSE, "PNG: truncated chunk");
        const uint8_t* body = data + off + 8;
        uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4);
        if (crc != be32(body + len)) throw Error(PT_ERR_PARSE, "PNG: chunk CRC mismatch");
        if (memcmp(type, "IHDR", 4) == 0) {
            if (len != 13) throw Error(PT_ERR_PARSE, "PNG: bad IHDR");
            width = be32(body);
            height = be32(body + 4);
            bit_depth = body[8];
            color_type = body[9];
            interlace = body[12];
            if (body[10] != 0 || body[11] != 0) throw Error(PT_ERR_PARSE, "PNG: unknown compression/filter method");
            have_ihdr = true;
        } else if (memcmp(type, "PLTE", 4) == 0) {
            palette.assign(body, body + len);
        } else if (memcmp(type, "tRNS", 4) == 0) {
            trns.assign(body, body + len);
        } else if (memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), body, body + len);
        } else if (memcmp(type, "IEND", 4) == 0) {
            have_iend = true;
        }
        off += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_iend) throw Error(PT_ERR_PARSE, "PNG: missing IHDR or IEND");
    if (width == 0 || height == 0 || width > 65536 || height > 65536)
        throw Error(PT_ERR_UNSUPPORTED, "PNG: unsupported dimensions");
    if (interlace != 0) throw Error(PT_ERR_UNSUPPORTED, "PNG: interlaced images are not supported");
    int channels;
    switch (color_type) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw Error(PT_ERR_PARSE, "PNG: bad colour type");
    }
    bool depth_ok = (color_type == 3) ? (bit_depth == 1 || bit_depth == 2 || bit_depth == 4 || bit_depth == 8)
                                      : (bit_depth == 8);
    if (!depth_ok) throw Error(PT_ERR_UNSUPPORTED, "PNG: unsupported bit depth");
    if (color_type == 3 && (palette.empty() || palette.size() % 3 != 0)) throw Error(PT_ERR_PARSE, "PNG: bad PLTE");

    size_t bits_pp = (size_t)channels * bit_depth;
    size_t stride = ((size_t)width * bits_pp + 7) / 8;
    size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1; // filter byte distance
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) throw Error(PT_ERR_PARSE, "PNG: zlib init failed");
        zs.next_in = idat.data();
        zs.avail_in = (uInt)idat.size();
        size_t produced = 0;
        int rc = Z_OK;
        while (rc != Z_STREAM_END) {
            size_t room = raw.size() - produced;
            if (room == 0) break;
            uInt chunk = room > (1u << 30) ? (1u << 30) : (uInt)room;
            zs.next_out = raw.data() + produced;
            zs.avail_out = chunk;
            rc = inflate(&zs, Z_NO_FLUSH);
            produced += chunk - zs.avail_out;
            if (rc != Z_OK && rc != Z_STREAM_END) {
                inflateEnd(&zs);
                throw Error(PT_ERR_PARSE, "PNG: corrupt IDAT stream");
            }
            if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) break;
        }
        inflateEnd(&zs);
        if (produced != raw.size()) throw Error(PT_ERR_PARSE, "PNG: IDAT stream has the wrong length");
    }

    // undo the per-scanline filters in place
    for (uint32_t y = 0; y < height; y++) {
        uint8_t* line = raw.data() + (size_t)y * (stride + 1);
        int filter = line[0];
        uint8_t* cur = line + 1;
        const uint8_t* prev = y ? cur - (stride + 1) : nullptr;
        switch (filter) {
        case 0: break;
        case 1:
            for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            if (prev)
                for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + prev[i]);
            break;
        case 3:
            for (size_t i = 0; i < stride; i++) {
                int a = i >= bpp ? cur[i - bpp] : 0;
                int b = prev ? prev[i] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < stride; i++) {
                int a = i >= bpp ? cur[i - bpp] : 0;
                int b = prev ? prev[i] : 0;
                int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
            }
            break;
        default: throw Error(PT_ERR_PARSE, "PNG: unknown scanline filter");
        }
    }

    HostImage img;
    img.width = width;
    img.height = height;
    img.rgba8.resize((size_t)width * height * 4);
    size_t pal_entries = palette.size() / 3;
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t* cur = raw.data() + (size_t)y * (stride + 1) + 1;
        uint8_t* out = img.rgba8.data() + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; x++, out += 4) {
            switch (color_type) {
            case 0:
                out[0] = out[1] = out[2] = cur[x];
                out[3] = 255;
                break;
            case 2:
                out[0] = cur[3 * x];
                out[1] = cur[3 * x + 1];
                out[2] = cur[3 * x + 2];
                out[3] = 255;
                break;
            case 4:
                out[0] = out[1] = out[2] = cur[2 * x];
                out[3] = cur[2 * x + 1];
                break;
            case 6:
                memcpy(out, cur + 4 * (size_t)x, 4);
                break;
            case 3: {
                size_t idx;
                if (bit_depth == 8) {
                    idx = cur[x];
                } else {
                    size_t bit = (size_t)x * bit_depth;
                    idx = (cur[bit >> 3] >> (8 - bit_depth - (bit & 7))) & ((1u << bit_depth) - 1);
                }
                if (idx >= pal_entries) throw Error(PT_ERR_PARSE, "PNG: palette index out of range");
                out[0] = palette[3 * idx];
                out[1] = palette[3 * idx + 1];
                out[2] = palette[3 * idx + 2];
                out[3] = idx < trns.size() ? trns[idx] : 255;
                break;
            }
            }
        }
    }
    return img;
}

} // namespace pt
