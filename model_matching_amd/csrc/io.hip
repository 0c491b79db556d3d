// io.hip -- file formats either side of the hot path (host code only; SURVEY.md 8f-4 "pose-file / PLY writers" and the
// readers the reference-shaped constructor needs).  Replaces, for the files the reference's driver touches,
//   cv::imread of depth / class-probability / edge PNGs        reference src/rgbd.cpp:197-199, src/stocs.cpp:117
//   pcl::io::loadPLYFile / rgbd::save_as_ply                   reference src/stocs.cpp:45,91, src/rgbd.cpp:35-56
// with self-contained code (zlib is the only dependency): OpenCV, PCL and libpng are absent from this image.
// PNG: 8/16-bit grey, grey+alpha, RGB, RGBA, non-interlaced (what the reference's example data uses); 16-bit samples are
// returned in host byte order.  PLY: ascii or binary_little_endian, float x y z with optional normal_x/nx ... properties.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

#include "stocs_ctx.h"

namespace stocs {

static uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

static bool read_file(const char* path, std::vector<unsigned char>* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out->resize((size_t)n);
    const bool ok = n == 0 || fread(out->data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

static inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// decodes into `pix` (rows of width*channels samples, 1 or 2 bytes each, 16-bit in host order)
static int png_decode(const char* path, int* W, int* H, int* C, int* bits, std::vector<unsigned char>* pix) {
    std::vector<unsigned char> file;
    if (!read_file(path, &file)) { set_error("cannot read %s", path); return STOCS_ERR_INVALID; }
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) { set_error("%s is not a PNG file", path); return STOCS_ERR_INVALID; }
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat;
    bool end = false;
    while (!end && pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) { set_error("%s: truncated chunk", path); return STOCS_ERR_INVALID; }
        const unsigned char* data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    int ch = 0;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 4: ch = 2; break; case 6: ch = 4; break; default: ch = 0; }
    if (!w || !h || w > 32768 || h > 32768 || ch == 0 || (depth != 8 && depth != 16) || interlace != 0) {
        set_error("%s: unsupported PNG (colour type %d, bit depth %d, interlace %d); 8/16-bit grey / RGB (+alpha), non-interlaced are handled", path, ctype, depth, interlace);
        return STOCS_ERR_INVALID;
    }
    const size_t bps = (size_t)depth / 8, bpp = bps * ch, stride = (size_t)w * bpp;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) {
        set_error("%s: corrupt image data", path);
        return STOCS_ERR_INVALID;
    }
    pix->assign(stride * (size_t)h, 0);
    for (uint32_t y = 0; y < h; ++y) {   // undo the per-scanline filter (PNG specification, section 9)
        const unsigned char* src = &raw[(stride + 1) * (size_t)y];
        unsigned char* dst = &(*pix)[stride * (size_t)y];
        const unsigned char* up = y ? dst - stride : NULL;
        const int ft = src[0];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? dst[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = src[1 + x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: set_error("%s: bad filter type %d", path, ft); return STOCS_ERR_INVALID;
            }
            dst[x] = (unsigned char)v;
        }
    }
    if (depth == 16)   // big endian on file -> host order
        for (size_t i = 0; i + 1 < pix->size(); i += 2) { const uint16_t v = (uint16_t)(((*pix)[i] << 8) | (*pix)[i + 1]); memcpy(&(*pix)[i], &v, 2); }
    *W = (int)w; *H = (int)h; *C = ch; *bits = depth;
    return STOCS_OK;
}

struct PlyProp { std::string type, name; };
static int ply_type_size(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
static double ply_bin_value(const unsigned char* p, const std::string& t) {
    if (t == "float" || t == "float32") { float v; memcpy(&v, p, 4); return v; }
    if (t == "double" || t == "float64") { double v; memcpy(&v, p, 8); return v; }
    if (t == "uchar" || t == "uint8") return *p;
    if (t == "char" || t == "int8") return (signed char)*p;
    if (t == "short" || t == "int16") { int16_t v; memcpy(&v, p, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, p, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; memcpy(&v, p, 4); return v; }
    uint32_t v; memcpy(&v, p, 4); return v;
}

}  // namespace stocs

using namespace stocs;

extern "C" {

int stocs_png_read(const char* path, int* width, int* height, int* channels, int* bit_depth, void* pixels, int64_t cap_bytes) {
    if (!path || !width || !height || !channels || !bit_depth) return STOCS_ERR_INVALID;
    std::vector<unsigned char> pix;
    const int rc = png_decode(path, width, height, channels, bit_depth, &pix);
    if (rc) return rc;
    if (!pixels) return STOCS_OK;   // size query
    if ((int64_t)pix.size() > cap_bytes) { set_error("stocs_png_read: %zu bytes needed, capacity %lld", pix.size(), (long long)cap_bytes); return STOCS_ERR_CAPACITY; }
    memcpy(pixels, pix.data(), pix.size());
    return STOCS_OK;
}

int stocs_ply_read(const char* path, float* pos3, float* nrm3, int cap, int* n, int* has_normals) {
    if (!path || !n) return STOCS_ERR_INVALID;
    std::vector<unsigned char> file;
    if (!read_file(path, &file)) { set_error("cannot read %s", path); return STOCS_ERR_INVALID; }
    // header: lines up to "end_header"
    size_t pos = 0;
    auto next_line = [&](std::string* line) {
        if (pos >= file.size()) return false;
        size_t e = pos;
        while (e < file.size() && file[e] != '\n') ++e;
        line->assign((const char*)&file[pos], e - pos);
        if (!line->empty() && line->back() == '\r') line->pop_back();
        pos = e + 1;
        return true;
    };
    std::string line;
    if (!next_line(&line) || line != "ply") { set_error("%s is not a PLY file", path); return STOCS_ERR_INVALID; }
    int fmt = -1;   // 0 ascii, 1 binary little endian
    long long nv = -1;
    bool in_vertex = false, other_element_first = false;
    std::vector<PlyProp> props;
    bool ended = false;
    while (next_line(&line)) {
        char a[64] = "", b[64] = "", c3[64] = "";
        const int k = sscanf(line.c_str(), "%63s %63s %63s", a, b, c3);
        if (k >= 1 && !strcmp(a, "end_header")) { ended = true; break; }
        if (k >= 2 && !strcmp(a, "format")) fmt = !strcmp(b, "ascii") ? 0 : (!strcmp(b, "binary_little_endian") ? 1 : -1);
        else if (k >= 3 && !strcmp(a, "element")) {
            in_vertex = !strcmp(b, "vertex");
            if (in_vertex) nv = atoll(c3);
            else if (nv < 0) other_element_first = true;
        } else if (k >= 3 && !strcmp(a, "property") && in_vertex) {
            if (!strcmp(b, "list")) { set_error("%s: list property in the vertex element", path); return STOCS_ERR_INVALID; }
            PlyProp p; p.type = b; p.name = c3; props.push_back(p);
        }
    }
    if (!ended || fmt < 0 || nv < 0 || other_element_first) { set_error("%s: unsupported PLY header (ascii / binary_little_endian with the vertex element first)", path); return STOCS_ERR_INVALID; }
    int ix[6] = {-1, -1, -1, -1, -1, -1};
    for (size_t i = 0; i < props.size(); ++i) {
        const std::string& nm = props[i].name;
        if (nm == "x") ix[0] = (int)i; else if (nm == "y") ix[1] = (int)i; else if (nm == "z") ix[2] = (int)i;
        else if (nm == "normal_x" || nm == "nx") ix[3] = (int)i; else if (nm == "normal_y" || nm == "ny") ix[4] = (int)i;
        else if (nm == "normal_z" || nm == "nz") ix[5] = (int)i;
    }
    if (ix[0] < 0 || ix[1] < 0 || ix[2] < 0) { set_error("%s: no x / y / z vertex properties", path); return STOCS_ERR_INVALID; }
    const bool hn = ix[3] >= 0 && ix[4] >= 0 && ix[5] >= 0;
    if (has_normals) *has_normals = hn ? 1 : 0;
    *n = (int)nv;
    if (!pos3) return STOCS_OK;   // size query
    if (nv > cap) { set_error("stocs_ply_read: %lld vertices, capacity %d", nv, cap); return STOCS_ERR_CAPACITY; }
    std::vector<double> vals(props.size());
    size_t rec = 0;
    std::vector<size_t> off(props.size());
    for (size_t i = 0; i < props.size(); ++i) { off[i] = rec; const int sz = ply_type_size(props[i].type); if (!sz) { set_error("%s: unknown property type %s", path, props[i].type.c_str()); return STOCS_ERR_INVALID; } rec += (size_t)sz; }
    for (long long v = 0; v < nv; ++v) {
        if (fmt == 0) {
            if (!next_line(&line)) { set_error("%s: truncated vertex list", path); return STOCS_ERR_INVALID; }
            const char* s = line.c_str();
            for (size_t i = 0; i < props.size(); ++i) {
                char* e = NULL;
                vals[i] = strtod(s, &e);
                if (e == s) { set_error("%s: bad vertex line %lld", path, v); return STOCS_ERR_INVALID; }
                s = e;
            }
        } else {
            if (pos + rec > file.size()) { set_error("%s: truncated vertex data", path); return STOCS_ERR_INVALID; }
            for (size_t i = 0; i < props.size(); ++i) vals[i] = ply_bin_value(&file[pos + off[i]], props[i].type);
            pos += rec;
        }
        for (int k = 0; k < 3; ++k) pos3[3 * v + k] = (float)vals[(size_t)ix[k]];
        if (nrm3) for (int k = 0; k < 3; ++k) nrm3[3 * v + k] = hn ? (float)vals[(size_t)ix[3 + k]] : 0.0f;
    }
    return STOCS_OK;
}

// rgbd::save_as_ply (rgbd.cpp:35-56): positions scaled, normals as they are; ascii, 9 significant digits (floats round-trip)
int stocs_ply_write(const char* path, const float* pos3, const float* nrm3, int n, float scale) {
    if (!path || n < 0 || (n && !pos3)) return STOCS_ERR_INVALID;
    FILE* f = fopen(path, "w");
    if (!f) { set_error("cannot write %s", path); return STOCS_ERR_INVALID; }
    fprintf(f, "ply\nformat ascii 1.0\ncomment written by libstocs_hip\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n", n);
    if (nrm3) fprintf(f, "property float normal_x\nproperty float normal_y\nproperty float normal_z\n");
    fprintf(f, "end_header\n");
    for (int i = 0; i < n; ++i) {
        fprintf(f, "%.9g %.9g %.9g", (double)(pos3[3 * i] * scale), (double)(pos3[3 * i + 1] * scale), (double)(pos3[3 * i + 2] * scale));
        if (nrm3) fprintf(f, " %.9g %.9g %.9g", (double)nrm3[3 * i], (double)nrm3[3 * i + 1], (double)nrm3[3 * i + 2]);
        fputc('\n', f);
    }
    const bool ok = !ferror(f);
    return (fclose(f) == 0 && ok) ? STOCS_OK : STOCS_ERR_INVALID;
}

}  // extern "C"
