// mzd_k_headers.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
// ------------------------------------------------------------------------------------ K0 + block driver
__device__ __noinline__ void parse_frame_or_skip(Ctl& c, const uint8_t* src, uint64_t n, const DevDict* dicts, uint32_t ndicts, uint32_t job_dict) {
    MZD_IN_LDS(&c);
    uint64_t pos = c.pos;
    if (pos >= n) { c.action = 2; return; }
    if (n - pos < 4) { c.err = MZD_E_TRUNCATED; return; }
    const uint8_t* p = src + pos;
    uint32_t magic = ld32(p);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
        if (n - pos < 8) { c.err = MZD_E_TRUNCATED; return; }
        uint64_t sz = ld32(p + 4);
        if (n - pos - 8 < sz) { c.err = MZD_E_TRUNCATED; return; }
        c.pos = pos + 8 + sz;
        c.action = 1;
        return;
    }
    if (magic != 0xFD2FB528u) { c.err = MZD_E_BADMAGIC; return; }
    if (n - pos < 5) { c.err = MZD_E_TRUNCATED; return; }
    uint32_t fhd = p[4];
    uint32_t fcsf = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    if (fhd & 0x08) { c.err = MZD_E_UNSUPPORTED; return; }
    uint32_t did_sz = did == 3 ? 4 : did, fcs_sz = fcsf == 0 ? single : (1u << fcsf);
    uint64_t hs = 5 + (single ? 0 : 1) + did_sz + fcs_sz;
    if (n - pos < hs) { c.err = MZD_E_TRUNCATED; return; }
    const uint8_t* q = p + 5;
    uint64_t window = 0;
    if (!single) { uint32_t b = *q++; uint32_t wl = 10 + (b >> 3); window = (1ull << wl) + ((1ull << wl) >> 3) * (b & 7); }
    uint32_t dict_id = 0;
    if (did == 1) { dict_id = q[0]; q += 1; } else if (did == 2) { dict_id = ld16(q); q += 2; } else if (did == 3) { dict_id = ld32(q); q += 4; }
    c.has_fcs = 1;
    if (fcsf == 0) { if (single) c.fcs = *q++; else { c.fcs = 0; c.has_fcs = 0; } }
    else if (fcsf == 1) { c.fcs = (uint64_t)ld16(q) + 256; }
    else if (fcsf == 2) { c.fcs = ld32(q); }
    else { c.fcs = ld64(q); }
    if (single) window = c.fcs;
    if (window > (1ull << 27) + 1) { c.err = MZD_E_UNSUPPORTED; return; } // copy_decode is a streaming decoder (windowLogMax 27)
    c.block_max = (uint32_t)(window < kBlockMax ? window : kBlockMax);
    c.has_cksum = (fhd >> 2) & 1;
    c.pos = pos + hs;
    c.frame_out0 = c.out;
    c.rep[0] = 1; c.rep[1] = 4; c.rep[2] = 8;
    c.huf_valid = 0; c.fse_valid = 0;
    c.dict_content = nullptr; c.dict_content_len = 0;
    c.action = 0;
    // dictionary
    const DevDict* dd = (job_dict >= 1 && job_dict <= ndicts) ? &dicts[job_dict - 1] : nullptr;
    // libzstd: a frame that names a dictionary fails unless exactly that dictionary is loaded
    if (dict_id && dict_id != (dd && dd->formatted ? dd->dict_id : 0u)) { c.err = MZD_E_DICT; return; }
    if (dd) c.action = 3; // frame with dictionary: tables are copied in by the workgroup
}

__device__ __noinline__ void parse_block_header(Ctl& c, const uint8_t* src, uint64_t n) {
    MZD_IN_LDS(&c);
    if (n - c.pos < 3) { c.err = MZD_E_TRUNCATED; return; }
    uint32_t bh = ld24(src + c.pos);
    c.pos += 3;
    c.last = bh & 1; c.btype = (bh >> 1) & 3; c.bsize = bh >> 3;
    if (c.btype == 3 || c.bsize > c.block_max) { c.err = MZD_E_CORRUPT; return; }
    uint64_t need = c.btype == 1 ? 1 : c.bsize;
    if (n - c.pos < need) { c.err = MZD_E_TRUNCATED; return; }
    if (c.btype == 2 && c.bsize < 2) { c.err = MZD_E_CORRUPT; return; }
}

// literals section header (+ Huffman weights).  Lane 0.
__device__ __noinline__ void parse_literals(Ctl& c, const uint8_t* b, uint32_t n) {
    MZD_IN_LDS(&c); MZD_IN_LDS(b);
    uint32_t type = b[0] & 3, sf = (b[0] >> 2) & 3;
    uint32_t regen, comp = 0, hs, streams = 0;
    c.lit_type = type;
    c.lit_is_raw = 0;
    if (type < 2) {
        if (sf == 0 || sf == 2) { hs = 1; regen = b[0] >> 3; }
        else if (sf == 1) { if (n < 2) { c.err = MZD_E_CORRUPT; return; } hs = 2; regen = (b[0] >> 4) + ((uint32_t)b[1] << 4); }
        else { if (n < 3) { c.err = MZD_E_CORRUPT; return; } hs = 3; regen = (b[0] >> 4) + ((uint32_t)b[1] << 4) + ((uint32_t)b[2] << 12); }
        if (regen > c.block_max) { c.err = MZD_E_CORRUPT; return; }
        uint32_t body = type == 0 ? regen : 1;
        if (hs + body > n) { c.err = MZD_E_CORRUPT; return; }
        c.nlit = regen; c.streams = 0;
        c.lit_off = c.pos + hs;
        c.lit_is_raw = type == 0;
        c.seq_off = c.pos + hs + body;
        c.seq_len = n - hs - body;
        return;
    }
    if (n < 3) { c.err = MZD_E_CORRUPT; return; }
    if (type == 3 && !c.huf_valid) { c.err = MZD_E_DICT; return; } // treeless literals without a tree: libzstd's dictionary_corrupted, found before the section's sizes are looked at
    if (sf == 0 || sf == 1) { hs = 3; uint32_t v = ld24(b); regen = (v >> 4) & 0x3FF; comp = v >> 14; streams = sf ? 4 : 1; }
    else if (sf == 2) { if (n < 4) { c.err = MZD_E_CORRUPT; return; } hs = 4; uint32_t v = ld32(b); regen = (v >> 4) & 0x3FFF; comp = v >> 18; streams = 4; }
    else { if (n < 5) { c.err = MZD_E_CORRUPT; return; } hs = 5; uint64_t v = (uint64_t)ld32(b) | ((uint64_t)b[4] << 32); regen = (uint32_t)(v >> 4) & 0x3FFFF; comp = (uint32_t)(v >> 22); streams = 4; }
    if (regen > c.block_max || regen == 0 || (streams == 4 && regen < 6) || hs + comp > n) { c.err = MZD_E_CORRUPT; return; }
    const uint8_t* p = b + hs;
    uint32_t rem = comp;
    if (type == 2) { // the tree is decoded later by another wavefront; here only its extent
        if (rem < 1) { c.err = MZD_E_CORRUPT; return; }
        uint32_t hb = p[0];
        uint32_t tl = hb >= 128 ? 1 + ((hb - 127) + 1) / 2 : 1 + hb;
        if (tl > rem) { c.err = MZD_E_CORRUPT; return; }
        c.huf_tree_off = (uint32_t)(p - b); c.huf_tree_len = tl;
        p += tl; rem -= tl;
    }
    uint32_t base = (uint32_t)(p - b); // offset of the streams inside the block
    if (streams == 1) {
        c.s_off[0] = base; c.s_len[0] = rem; c.s_out[0] = 0; c.s_n[0] = regen;
    } else {
        if (rem < 10) { c.err = MZD_E_CORRUPT; return; }
        uint32_t l1 = ld16(p), l2 = ld16(p + 2), l3 = ld16(p + 4);
        if (6 + l1 + l2 + l3 > rem) { c.err = MZD_E_CORRUPT; return; }
        uint32_t l4 = rem - 6 - l1 - l2 - l3;
        uint32_t seg = (regen + 3) / 4;
        if (3 * seg > regen) { c.err = MZD_E_CORRUPT; return; }
        c.s_off[0] = base + 6; c.s_off[1] = c.s_off[0] + l1; c.s_off[2] = c.s_off[1] + l2; c.s_off[3] = c.s_off[2] + l3;
        c.s_len[0] = l1; c.s_len[1] = l2; c.s_len[2] = l3; c.s_len[3] = l4;
        c.s_out[0] = 0; c.s_out[1] = seg; c.s_out[2] = 2 * seg; c.s_out[3] = 3 * seg;
        c.s_n[0] = c.s_n[1] = c.s_n[2] = seg; c.s_n[3] = regen - 3 * seg;
    }
    c.nlit = regen; c.streams = streams;
    c.seq_off = c.pos + hs + comp;
    c.seq_len = n - hs - comp;
}

// sequences section header: nbSeq, modes, table descriptions.  Lane 0 of the walking wavefront, while other
// wavefronts already work on the literals (errors are posted first-wins).
// `stage_off`: where `b` lies inside S.stage (the normalized-count reader addresses the staging area by offset)
__device__ __noinline__ void parse_seq_header(Ctl& c, const uint8_t* b, uint32_t n, uint32_t stage_off) {
    MZD_IN_LDS(&c);
    if (n < 1) { post_err(&c.err, MZD_E_CORRUPT); return; }
    const uint8_t* p = b;
    const uint8_t* end = b + n;
    uint32_t nseq = *p++;
    if (nseq > 0x7F) {
        if (nseq == 0xFF) { if (p + 2 > end) { post_err(&c.err, MZD_E_CORRUPT); return; } nseq = ld16(p) + 0x7F00; p += 2; }
        else { if (p + 1 > end) { post_err(&c.err, MZD_E_CORRUPT); return; } nseq = ((nseq - 0x80) << 8) + *p++; }
    }
    c.nseq = nseq;
    if (nseq == 0) { if (p != end) post_err(&c.err, MZD_E_CORRUPT); return; }
    if (nseq > kMaxSeq - 1 || p + 1 > end) { post_err(&c.err, MZD_E_CORRUPT); return; }
    uint32_t modes = *p++;
    if (modes & 3) { post_err(&c.err, MZD_E_CORRUPT); return; }
    c.mode[0] = modes >> 6; c.mode[1] = (modes >> 4) & 3; c.mode[2] = (modes >> 2) & 3;
    const int max_log[3] = {9, 8, 9}, max_sym[3] = {35, 31, 52};
    for (int t = 0; t < 3; t++) {
        uint32_t m = c.mode[t];
        if (m == 1) {
            if (p + 1 > end || *p > max_sym[t]) { post_err(&c.err, MZD_E_CORRUPT); return; }
            c.nsym[t] = *p++; // the symbol itself
        } else if (m == 2) {
            // the header was staged at S.stage + 256 by the caller
            const uint32_t at = (stage_off & ~kInRing) + (uint32_t)(p - b);
            int used = (stage_off & kInRing) ? read_ncount_ring(at, (uint32_t)(end - p), max_log[t], max_sym[t], S.norm[t], &c.nsym[t], &c.al[t])
                                             : read_ncount_staged(at, (uint32_t)(end - p), max_log[t], max_sym[t], S.norm[t], &c.nsym[t], &c.al[t]);
            if (used <= 0) { post_err(&c.err, MZD_E_CORRUPT); return; }
            p += used;
        } else if (m == 3) {
            if (!c.fse_valid) { post_err(&c.err, MZD_E_CORRUPT); return; }
        }
    }
    c.seq_off += (uint64_t)(p - b);
    c.seq_len = (uint32_t)(end - p);
}

// The three sequence tables of a block, built one after the other by ONE wavefront.
__device__ __noinline__ void build_tables_wave(int lane) {
    Ctl& c = S.c;
    for (int t = 0; t < 3; t++) {
        uint64_t* tab = t == 0 ? S.ll : (t == 1 ? S.of : S.ml);
        const uint32_t m = c.mode[t];
        if (m == 0) {
            const int16_t* def = t == 0 ? LL_DEF : (t == 1 ? OF_DEF : ML_DEF);
            const uint32_t n = t == 0 ? 36 : (t == 1 ? 29 : 53), lg = t == 1 ? 5 : 6;
            if ((uint32_t)lane < n) S.norm[t][lane] = def[lane];
            build_seq_table_wave(tab, S.norm[t], n, lg, t, S.ring, lane);
            if (lane == 0) c.al[t] = lg;
        } else if (m == 1) {
            if (lane == 0) { rle_seq_table(tab, c.nsym[t], t); c.al[t] = 0; }
        } else if (m == 2) {
            build_seq_table_wave(tab, S.norm[t], c.nsym[t], c.al[t], t, S.ring, lane);
        }
    }
}

// Control words live in LDS and are written by lane 0 (or one lane per wavefront).  Every
// decision the workgroup takes on them is read through WG_SNAPSHOT: barrier, every lane copies
// the words it needs into registers, barrier -- so no lane can still be reading a word when the
// next step rewrites it, and all 256 lanes always take the same branch.
#define WG_SNAPSHOT(...) do { grp_sync(); __VA_ARGS__; grp_sync(); } while (0)


// The launch's queue: tickets are job indices, or -- behind the small-file kernel -- indices into the launch's job list
// (the host's part, then what that kernel handed on: KernelArgs::job_list).
__device__ __forceinline__ uint32_t queue_len(const KernelArgs& a) { return a.job_list ? a.nlist_fixed + __atomic_load_n(&a.counter[4], __ATOMIC_RELAXED) : a.njobs; }
__device__ __forceinline__ uint32_t queue_job(const KernelArgs& a, uint32_t ticket) { return a.job_list ? a.job_list[ticket] : ticket; }
// Tickets: a workgroup's FIRST ticket is its own index -- a thousand workgroups starting at once would otherwise queue up on one
// atomic counter (~30 K cycles at the median) -- and the later ones come from the counter, which therefore counts from the number of groups.
__device__ __forceinline__ uint32_t take_ticket(const KernelArgs& a) { // thread 0
    if (!S.took_first) { S.took_first = 1; return vblock(); }
    return vgrid() + atomicAdd(&a.counter[0], 1u);
}
__device__ __forceinline__ uint32_t take_job(const KernelArgs& a) { // thread 0
    for (;;) {
        const uint32_t t = take_ticket(a);
        if (t >= queue_len(a)) return kDoneJob;
        const uint32_t j = queue_job(a, t);
        if (j < a.njobs) return j; // (a list entry that names no job of the launch is skipped, never decoded: the list is the host's or the small-file kernel's)
    }
}

// Driver 1, by the walking wavefront once its own work on a file's last block is done: take the next file and parse
// the headers of its first block (frame header, block header, literals header, sequence header with its three
// normalized-count descriptions: ~60 K cycles of serial parsing) into S.c2, so that the workgroup finds them ready
// when the copier and the hasher are through with the current file.  Only the plain case is prepared (one frame start,
// a compressed first block, no error); anything else leaves pre_valid = 0 and the file is parsed the normal way.
__device__ __noinline__ void pre_parse_next(const KernelArgs& a, int lane) {
    Ctl& c2 = S.c2;
    uint32_t j2 = 0;
    if (lane == 0) { j2 = take_job(a); S.pre_job = j2; S.pre_valid = 0; }
    j2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)j2);
    if (j2 >= a.njobs) return;
    const uint8_t* const src = a.jobs[j2].src;
    const uint64_t n = a.jobs[j2].src_len;
    const uint32_t job_dict = a.jobs[j2].dict;
    if (lane == 0) { S.pj.src = src; S.pj.n = n; S.pj.dst = a.jobs[j2].dst; S.pj.cap = a.jobs[j2].dst_cap; S.pj.dict = job_dict; }
    if (lane == 0) {
        c2.pos = 0; c2.out = 0; c2.err = 0; c2.action = 0; c2.btype = 0; c2.diag_slow = 0;
        if (job_dict > a.ndicts) c2.err = MZD_E_DICT;
        else parse_frame_or_skip(c2, src, n, a.dicts, a.ndicts, job_dict);
        if (!c2.err && (c2.action == 0 || c2.action == 3)) {
            if (c2.action == 3 && a.dicts[job_dict - 1].formatted) { c2.huf_valid = 1; c2.fse_valid = 1; } // (the tables themselves are loaded by the workgroup)
            parse_block_header(c2, src, n);
        } else if (!c2.err) c2.err = MZD_E_PARAM; // skippable frame / end of file: not prepared
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (c2.err || c2.btype != 2) return; // (wave-uniform: every lane reads the same words)
    const uint64_t pos0 = c2.pos;
    const uint32_t bsize = c2.bsize;
    uint8_t* const ps = S.ring + kPreStage;
    for (uint32_t k = (uint32_t)lane; k < bsize && k < 256; k += 64) ps[k] = src[pos0 + k];
    if (lane == 0) parse_literals(c2, ps, bsize);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (c2.err) return;
    const uint64_t seq_off = c2.seq_off;
    const uint32_t seq_len = c2.seq_len;
    for (uint32_t k = (uint32_t)lane; k < seq_len && k < 256; k += 64) ps[256 + k] = src[seq_off + k];
    if (lane == 0) parse_seq_header(c2, ps + 256, seq_len, kInRing | (kPreStage + 256));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (c2.err) return;
    // (Building the Huffman table ahead as well was tried and measured slower: the ~65 K cycles of weight decoding then
    //  queue behind this wavefront's own walk instead of running beside it on the copying wavefront.)
    if (lane == 0) S.pre_valid = 1;
}

