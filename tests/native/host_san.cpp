// host_san.cpp -- TEST INFRASTRUCTURE.  Drives the product's HOST-side code (libmzd's frame / block header walks, the lazy open's
// index and its synthetic-frame builder, the open / read / release mirror) over untrusted bytes, in a build of the product with
// AddressSanitizer + UndefinedBehaviorSanitizer on the host side (`make -C fuse_zstd_amd/csrc asan`; the device side is compiled
// as it ships: GPU sanitizers do not exist on this pool).  No GPU is needed: every decode entry point answers MZD_E_DEVICE /
// -EFAULT without one, after the host code in front of it has run.  Inputs: the files named on the command line (golden
// vectors) -- each also in every truncation when it is short, in 64 truncations otherwise -- and hostile headers built here.
// Every input sits in a heap block of exactly its size, so a read past the end is an ASan report.  Exit code 0 = no report.
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mzd.h"

extern "C" int mzd_debug_lazy_plan(const uint8_t* zst, size_t n, uint32_t frame, uint32_t nblocks, uint8_t* synth, size_t cap, size_t* synth_len, uint64_t* total, uint32_t* nblocks_of_frame);

static unsigned long g_calls = 0;

static void one(const std::vector<uint8_t>& v) {
    uint8_t* p = (uint8_t*)malloc(v.size() ? v.size() : 1); // exactly its size (malloc(0) may return null)
    if (v.size()) memcpy(p, v.data(), v.size());
    const size_t n = v.size();
    const uint64_t cs = mzd_content_size(p, n);
    (void)cs;
    uint64_t total = 0;
    uint32_t nb = 0;
    size_t sl = 0;
    std::vector<uint8_t> synth(n + 64);
    const int nframes = mzd_debug_lazy_plan(p, n, 0, 1, synth.data(), synth.size(), &sl, &total, &nb);
    for (int f = 0; f < nframes && f < 8; f++) {
        uint32_t nbf = 0;
        mzd_debug_lazy_plan(p, n, (uint32_t)f, 1, synth.data(), synth.size(), &sl, &total, &nbf);
        for (uint32_t k = 1; k <= nbf && k <= 6; k++) {
            sl = 0;
            mzd_debug_lazy_plan(p, n, (uint32_t)f, k, synth.data(), synth.size(), &sl, &total, &nbf);
            if (sl > synth.size()) { fprintf(stderr, "synthetic frame longer than its source\n"); exit(3); }
            if (sl) { // what was built is itself a file the header walks accept up to its content size being unknown
                std::vector<uint8_t> s2(synth.begin(), synth.begin() + (ptrdiff_t)sl);
                uint8_t* q = (uint8_t*)malloc(sl);
                memcpy(q, s2.data(), sl);
                if (mzd_content_size(q, sl) == MZD_CONTENTSIZE_ERROR) { fprintf(stderr, "synthetic frame does not parse\n"); exit(3); }
                free(q);
            }
        }
    }
    mzd_fs* fs = mzd_fs_new();
    uint64_t rs = 0;
    const int64_t h1 = mzd_fs_open(fs, 7, 0, p, n, &rs);          // no GPU: -EFAULT after the header walk (or a handle for an empty file)
    const int64_t h2 = mzd_fs_open_lazy(fs, 8, 0, p, n, &rs);
    uint8_t out[64];
    if (h1 >= 0) { mzd_fs_read(fs, (uint64_t)h1, 0, 64, out); mzd_fs_release(fs, (uint64_t)h1); }
    if (h2 >= 0) { mzd_fs_read(fs, (uint64_t)h2, 0, 64, out); mzd_fs_release(fs, (uint64_t)h2); }
    if (mzd_fs_read(fs, 12345, 0, 64, out) != -ENOENT) { fprintf(stderr, "read of an unknown handle\n"); exit(3); }
    mzd_fs_free(fs);
    uint8_t dst[256];
    size_t ol = 0;
    (void)mzd_decode(p, n, dst, sizeof(dst), &ol); // MZD_E_DEVICE here; the entry's own checks run
    free(p);
    g_calls++;
}

static void with_truncations(const std::vector<uint8_t>& v) {
    one(v);
    const size_t n = v.size();
    if (n <= 600) { for (size_t k = 0; k < n; k++) one(std::vector<uint8_t>(v.begin(), v.begin() + (ptrdiff_t)k)); }
    else for (size_t i = 0; i < 64; i++) one(std::vector<uint8_t>(v.begin(), v.begin() + (ptrdiff_t)(n * i / 64)));
}

static void le(std::vector<uint8_t>& v, uint64_t x, int bytes) { for (int i = 0; i < bytes; i++) v.push_back((uint8_t)(x >> (8 * i))); }

int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
        FILE* f = fopen(argv[i], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[i]); return 2; }
        std::vector<uint8_t> v;
        uint8_t buf[65536];
        size_t r;
        while ((r = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + r);
        fclose(f);
        with_truncations(v);
    }
    // ---- hostile headers
    const uint64_t sizes[] = {1ull << 60, ~0ull, (1ull << 40) + 1, 1ull << 33, 0x7FFFFFFFFFFFFFFFull};
    for (uint64_t cs : sizes) {
        for (int single = 0; single < 2; single++) {
            std::vector<uint8_t> v;
            le(v, 0xFD2FB528u, 4);
            v.push_back((uint8_t)(0xC0 | (single << 5) | 4)); // 8-byte content size, checksum
            if (!single) v.push_back(0x70);                    // window descriptor
            le(v, cs, 8);
            le(v, 1 | (0 << 1) | (5 << 3), 3);                 // last raw block of 5 bytes
            le(v, 0x0102030405ull, 5);
            le(v, 0xDEADBEEF, 4);
            with_truncations(v);
        }
    }
    {   // a block that claims more bytes than the input holds; a reserved block type; a reserved header bit
        std::vector<uint8_t> v;
        le(v, 0xFD2FB528u, 4); v.push_back(0x20); v.push_back(10);
        le(v, 1 | (2 << 1) | (0x1FFFFFu << 3), 3);
        le(v, 0, 8);
        with_truncations(v);
        v[6] = (uint8_t)(1 | (3 << 1)); with_truncations(v);
        v[4] = 0x28; with_truncations(v);
    }
    {   // skippable frames: a size past the input, and sizes that wrap 32-bit arithmetic
        for (uint64_t sz : {0xFFFFFFFFull, 0xFFFFFFF8ull, 16ull, 0ull}) {
            std::vector<uint8_t> v;
            le(v, 0x184D2A53u, 4); le(v, sz, 4); le(v, 0x1122334455667788ull, 8);
            with_truncations(v);
        }
    }
    {   // a million empty frames (content size 0, one empty raw last block) and a million skippable frames
        std::vector<uint8_t> v;
        v.reserve(9000000);
        for (int i = 0; i < 1000000; i++) { le(v, 0xFD2FB528u, 4); v.push_back(0x20); v.push_back(0); le(v, 1, 3); }
        one(v);
        v.clear();
        for (int i = 0; i < 1000000; i++) { le(v, 0x184D2A50u, 4); le(v, 0, 4); }
        one(v);
    }
    {   // a frame of many tiny blocks (the lazy index keeps one entry per block)
        std::vector<uint8_t> v;
        le(v, 0xFD2FB528u, 4); v.push_back(0x80); v.push_back(0x50); le(v, 200000, 4);
        for (int i = 0; i < 200000; i++) { le(v, (i == 199999 ? 1u : 0u) | (1 << 1) | (1 << 3), 3); v.push_back((uint8_t)i); }
        one(v);
        with_truncations(std::vector<uint8_t>(v.begin(), v.begin() + 400));
    }
    printf("host_san: %lu inputs, no sanitizer report\n", g_calls);
    return 0;
}
