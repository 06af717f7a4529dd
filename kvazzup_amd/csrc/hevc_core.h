// kvazzup_amd/csrc/hevc_core.h -- per-thread building blocks of the HIP encoder/decoder that
// are serial by nature (CABAC, CU syntax, merge/AMVP signalling, intra sample prediction,
// deblocking of one edge segment).  Written as host+device inline functions so the same code
// runs inside the kernels (enc_kernels.hip / dec_kernels.hip) and can be unit-tested on the
// host (tests/hostcheck).  No HIP intrinsics in this file.
//
// The path being replaced: everything inside kvz_api->encoder_encode
// (/root/reference/src/media/processing/kvazaarfilter.cpp:435-438) and libOpenHevcDecode
// (/root/reference/src/media/processing/openhevcfilter.cpp:145-146).
#pragma once
#include "hevc_tables.h"

namespace kvzx {

KVZ_HD int iabs(int v) { return v < 0 ? -v : v; }
KVZ_HD int imin(int a, int b) { return a < b ? a : b; }
KVZ_HD int imax(int a, int b) { return a > b ? a : b; }
KVZ_HD int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
KVZ_HD int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
KVZ_HD int ilog2(unsigned v) { int n = 0; while (v > 1) { v >>= 1; n++; } return n; }
KVZ_HD int kv_clz32(uint32_t v) { return __builtin_clz(v); }            // v != 0
KVZ_HD int kv_ctz32(uint32_t v) { return __builtin_ctz(v); }            // v != 0

// ---------------------------------------------------------------------------------------------
// Frame state shared by all encoder kernels.  All pointers are device memory (or host memory in
// the host unit tests).  Planes have pitch = coded width (luma) / coded width / 2 (chroma);
// the coded size is a multiple of 64 (CTU), so every CTU is complete.
// Per-CU arrays are indexed per 8x8 luma block ("b8"), pitch b8w = cw / 8; every 8x8 block of
// a CU carries the CU's values.
// ---------------------------------------------------------------------------------------------
// sample adaptive offset parameters of one CTU (7.3.8.3): type 0 off, 1 band, 2 edge (Cr shares Cb's type and class);
// offset[c][k] = SaoOffsetVal[k + 1].  Same layout as the checker's orc_sao_params.
struct SaoParams {
  uint8_t type[3], eo_class[3], band_pos[3];
  int8_t offset[3][4];
};


// Scaling factors m[x][y] of 8.6.4.2 for the active scaling lists (7.4.5; host: decoder.hip build_scaling), one byte per coefficient, raster inside the block
// (index = the level word's position field): [6][16] 4x4 | [6][64] 8x8 | [6][256] 16x16 | [2][1024] 32x32; matrix = 3 * inter + plane (32x32: inter)
#define KVZ_SCALING_BYTES 4064
KVZ_HD int scaling_offset(int log2n, int plane, int inter)
{
  switch (log2n) {
    case 2: return (3 * inter + plane) * 16;
    case 3: return 96 + (3 * inter + plane) * 64;
    case 4: return 480 + (3 * inter + plane) * 256;
    default: return 2016 + inter * 1024;
  }
}


// rate control v2, device-side state (rc_kernels.hip; statement of record: rc_band_decide() / rc_picture_start() in oracle/hevc_enc.c).
// acc[g]: arrivals << 40 | level cost of group g's workgroups (ONE atomic per workgroup: the one that completes the count has the group's cost);
// decided: number of groups whose QP is final (bits 0..3; group 0 from the start) and the QP steps of groups 1.. (3 bits each, biased by 3) -- what a
// waiting workgroup needs in the one word it polls; cost_sofar: the levels priced so far.
// (`decided` is polled by every waiting workgroup and acc[g] takes an atomic from every workgroup of group g: each on a 128-byte line of its own --
// with all of them in one line the polls queued in front of the atomics and the launch took 130 us instead of 90.)
#define KVZ_RC_ACC_STRIDE 16
struct RcState {
  uint32_t ratio_q8, ratio_valid, cost_sofar; uint32_t cost[8], cost_valid[8];
  alignas(128) uint32_t decided;
  alignas(128) unsigned long long acc[8 * KVZ_RC_ACC_STRIDE];
};
struct EncFrame {
  int cw, ch, b8w, b8h;
  int qp, qpc, lambda_q4, range;
  int is_intra, poc;
  int wpp;
  int satd;                 // intra mode search: SATD (8x8 Hadamard) instead of SAD
  int me_early;             // me-early-termination: blocks that match the co-located reference block to within 64 * lambda_q4 are not searched
  int subme;                // fractional-sample refinement level 0..4 (k_subpel)
  const uint8_t *scaling;   // `scaling-list default`: KVZ_SCALING_BYTES scaling factors of the default lists (dec_frame.h scaling_offset), NULL: flat
  int lossless;             // `lossless`: cu_transquant_bypass in every coding unit -- levels = residual samples, reconstruction = source (no transform, no quantiser)
  int intra_chain;          // "intra-chain" (default on): blocks on a CTU's left edge / its above-right corner block choose among the modes that do not read the left CTU's below-left / the above-right CTU's samples
  int intra_p;              // "uvgx intra-in-P v1": intra coding units in P pictures (statement: oracle/hevc_enc.c me_block32); me_cost16 = k_me's inter cost of every 16x16 block (0: not searched)
  uint32_t *me_cost16;
  uint32_t *edge_col[3];    // k_intra_recon: per plane [CTU][S] the CTU's right column of reconstructed samples as self-validating words, sample | chain_gen << 8 (kernel_common.h IB_EDGE_R, IntraNeighbours)
  unsigned long long *edge_row[3];   // ... and per plane [CTU][S / 4] the CTU's bottom row: words of four samples | chain_gen << 32
  uint32_t chain_gen;       // generation of this launch of the intra chain (1 .. 2^24 - 1; the arrays are cleared when it wraps)
  int analyse_alone;        // 1: nothing else is queued beside this intra picture's mode search (owf 0 / 1: one picture at a time) -- the search takes the whole chip instead of kAnalysePerCu workgroups per compute unit
  int chain_diags;          // how many anti-diagonals of workgroups k_intra_recon is launched with (0: the default, three; two when two pictures' chains run side by side)
  uint32_t *ip_arrive; uint64_t *ip_scratch;      // k_intra_analyse<P>: per listed block the number of its quarters' workgroups that are done (back to zero when the last has arrived) and their best (cost | mode << 32) per block size and position [listed block][20]
  uint32_t *me_cand;        // ... and the 32x32 blocks with a quarter above the gate: [0] their count (k_deblock_tile zeroes it for the next picture), [1 ..] their raster indices
  int rdoq, signhide;       // kvazaar rdoq / signhide: the level-adjustment pass behind the quantiser (adjust_group) and sign_data_hiding_enabled_flag
  int slices;               // 1: a slice segment ends with every CTU row, 2: with every tile (kvazaar slices=wpp / tiles): what k_tokenize closes a CTU with
  int mv_frame;             // mv-constraint: 0 none, 1 the displaced block stays inside the picture, 2 the same with a 4-sample margin on odd displacements
  int tile_rows;            // 1: no tiles; n: n tile rows, uniform spacing (6.5.1)
  int tile_cols;            // 1: full-width tiles; n: n tile columns, uniform spacing
  int row0, nrows;          // band of CTU rows the encoder kernels work on (nrows == 0: the whole picture); a band starts and ends on tile boundaries
  int chp;                  // ch | tile rows << 20 | tile columns << 26: the `ch` argument of avail64() and of everything that forwards to it
  const uint8_t *src[3];
  uint8_t *rec[3];
  const uint8_t *ref[3];
  const uint8_t *me_ref;    // the luma plane k_me searches: ref[0], or (me-source) the previous picture's padded source plane
  int16_t *coef[3];
  uint8_t *cu_log2, *cu_intra, *cu_flags, *cu_merge_idx, *cu_mvp_idx, *cu_intra_mode, *cu_cbf;
  int16_t *cu_mv, *cu_mvd;      // [b8][2]
  // intra analysis scratch
  uint8_t *im8, *im16, *im32; uint32_t *ic8, *ic16, *ic32;
  // entropy coding, GPU half: per-CTU token slots (tok_cap tokens each) filled by 16 units per CTU
  // (tok_cursor: tokens used per slot, tok_seg: [ctu][unit][piece] {offset, length}), the
  // prefix sum of the per-CTU counts and the dense z-ordered copy the host arithmetic coder reads
  // (tok_dense / tok_count_out live in host-mapped pinned memory)
  uint32_t *tok_list;           // P pictures (whole pictures): the (16x16 unit, role) pairs that have something to say -- [0] their count, [1 ..] unit << 2 | role; written by k_inter_signal, worked off by k_tokenize's list form, count zeroed by k_tok_compact.  NULL: every (unit, role) gets a wave
  uint16_t *tok_buf; int tok_cap; uint32_t *tok_cursor; uint32_t *tok_seg; uint32_t *tok_cursor_next;   // tok_cursor_next: the NEXT picture's cursors (two arrays take turns), zeroed by this picture's k_tok_compact
  uint16_t *tok_dense; uint32_t tok_dense_cap; int32_t *tok_count_out; uint32_t *tok_off_out; uint32_t *err_out;   // host-mapped pinned (host arithmetic coder) or, but for err_out, device memory (k_cabac_rows)
  uint32_t *ent_cursors;        // k_cabac_rows' two output cursors (NULL with the host coder): k_tok_compact zeroes them for the launch that follows
  // Per-CTU QP (cu_qp_delta_enabled_flag, quantisation group = CTU; NULL pointers = one QP per picture):
  //   ctu_qt: QP the CTU's blocks are quantised / dequantised with; ctu_qy: QpY of its CUs from the first one with residual
  //   on (8.6.1), ctu_delta: the coded CuQpDeltaVal (QpY of the CUs before that one = ctu_qy - ctu_delta), ctu_first: z index
  //   (8x8 units) of that first CU, 64 = none
  const int8_t *ctu_qt; int8_t *ctu_qy, *ctu_delta; uint8_t *ctu_first;
  // sample adaptive offset (NULL = off): rec[] is then the picture up to deblocking, sao_out[] the filtered picture that is
  // output and referenced; sao[] the per-CTU parameters (encoder: decided by k_sao, decoder: parsed)
  SaoParams *sao; uint8_t *sao_out[3];
  uint32_t *sync;               // [CTU][plane] progress counters (intra reconstruction wavefront: finished 8x8 units of the CTU)
  uint32_t *err;                // device-side error flags
  const uint32_t *intra_order;  // CTU (raster index) handled by the k-th workgroup triple of k_intra_recon: anti-diagonal wavefront order
  // rate control v2 (NULL = off): k_inter_recon reconstructs the CTU rows in rc_nb groups inside ONE launch -- a group's workgroups price their levels, the last
  // of them decides the next group's QP step, and that group's workgroups, prediction and forward transform done, wait for the decision in front of the quantiser
  // the head of a P picture's chain folded into its first kernel (k_me; pb_on): what k_picture_begin does for the pictures that start otherwise
  RcState *pb_rc; int8_t *pb_qt; const int8_t *pb_roi; uint32_t pb_bits3; int pb_nctu; int8_t pb_on, pb_slot3, pb_have3, pb_pad;
  RcState *rc; long long rc_target; int rc_nb, rc_slot;      // rc_target: bits for the picture; rc_slot: where the picture's level cost is filed (picture index & 7)
  unsigned long long *trace;    // KVAZZUP_AMD_INTRA_TRACE: per (CTU, plane) 8 words {start, first block, end, time in border waits, blocks, stores, publishes, number of blocks} of k_intra_recon, 100 MHz ticks; else NULL
};

enum { CU_SKIP = 1, CU_MERGE = 2 };

// quantiser QP of the CTU holding luma sample (x, y)
KVZ_HD int ctu_quant_qp(const EncFrame &f, int x, int y) { return f.ctu_qt ? f.ctu_qt[(y >> 6) * (f.cw >> 6) + (x >> 6)] : f.qp; }
// QpY of the CU holding luma sample (x, y) (deblocking, 8.7.2.5.3)
KVZ_HD int cu_qpy(const EncFrame &f, int x, int y)
{
  if (!f.ctu_qy) return f.qp;
  const int ctu = (y >> 6) * (f.cw >> 6) + (x >> 6), xi = (x & 63) >> 3, yi = (y & 63) >> 3;
  int z = 0;
  for (int b = 0; b < 3; b++) z |= ((xi >> b) & 1) << (2 * b) | ((yi >> b) & 1) << (2 * b + 1);
  return z >= f.ctu_first[ctu] ? f.ctu_qy[ctu] : f.ctu_qy[ctu] - f.ctu_delta[ctu];
}
KVZ_HD int band_rows(const EncFrame &f) { return f.nrows > 0 ? f.nrows : (f.ch >> 6); }
KVZ_HD int b8idx(const EncFrame &f, int x, int y) { return (y >> 3) * f.b8w + (x >> 3); }

// z-scan order address of the 4x4 block holding luma sample (x, y), CTU = 64 (H.265 6.5.2)
KVZ_HD uint32_t zaddr64(int x, int y, int w_ctbs)
{
  uint32_t ctb = (uint32_t)((y >> 6) * w_ctbs + (x >> 6));
  uint32_t xi = (uint32_t)(x & 63) >> 2, yi = (uint32_t)(y & 63) >> 2;
  xi = (xi | (xi << 2)) & 0x33u; xi = (xi | (xi << 1)) & 0x55u;      // spread the 4 bits of each coordinate
  yi = (yi | (yi << 2)) & 0x33u; yi = (yi | (yi << 1)) & 0x55u;
  return (ctb << 8) | xi | (yi << 1);
}
// ... for coding tree blocks of 1 << ctbl samples (the decoder: any of 64, 32, 16)
KVZ_HD uint32_t zaddr_ctb(int x, int y, int w_ctbs, int ctbl)
{
  const uint32_t ctb = (uint32_t)((y >> ctbl) * w_ctbs + (x >> ctbl)), m = (1u << ctbl) - 1u;
  uint32_t xi = ((uint32_t)x & m) >> 2, yi = ((uint32_t)y & m) >> 2;
  xi = (xi | (xi << 2)) & 0x33u; xi = (xi | (xi << 1)) & 0x55u;
  yi = (yi | (yi << 2)) & 0x33u; yi = (yi | (yi << 1)) & 0x55u;
  return (ctb << 8) | xi | (yi << 1);
}
// Tiles are full-width rows with uniform spacing: tile row i starts at CTB row (i * hc) / T (6.5.1).
KVZ_HD int tile_row_of(int hc, int T, int cy) { return ((cy + 1) * T - 1) / hc; }
KVZ_HD int tile_row_first(int hc, int T, int i) { return (i * hc) / T; }
KVZ_HD bool tile_row_starts_at(int hc, int T, int cy) { return T <= 1 ? cy == 0 : tile_row_first(hc, T, tile_row_of(hc, T, cy)) == cy; }
KVZ_HD bool tile_row_ends_at(int hc, int T, int cy) { return T <= 1 ? cy == hc - 1 : (cy == hc - 1 || tile_row_of(hc, T, cy + 1) != tile_row_of(hc, T, cy)); }
// ... and the same for columns
KVZ_HD int tile_col_of(int wc, int C, int cx) { return C <= 1 ? 0 : ((cx + 1) * C - 1) / wc; }
KVZ_HD int tile_col_first(int wc, int C, int j) { return C <= 1 ? (j ? wc : 0) : (j * wc) / C; }
KVZ_HD bool tile_col_starts_at(int wc, int C, int cx) { return C <= 1 ? cx == 0 : tile_col_first(wc, C, tile_col_of(wc, C, cx)) == cx; }
KVZ_HD bool tile_col_ends_at(int wc, int C, int cx) { return C <= 1 ? cx == wc - 1 : (cx == wc - 1 || tile_col_of(wc, C, cx + 1) != tile_col_of(wc, C, cx)); }
KVZ_HD int pack_height(int ch, int tile_rows, int tile_cols = 1) { return ch | ((tile_rows > 1 || tile_cols > 1 ? tile_rows : 0) << 20) | ((tile_cols > 1 ? tile_cols : 0) << 26); }
// H.265 6.4.1 for one slice; chp = coded height | tile rows << 20 | tile columns << 26 (pack_height): a neighbour in another tile is unavailable
KVZ_HD bool avail64(int cw, int chp, int xc, int yc, int xn, int yn)
{
  const int ch = chp & 0xfffff, T = (chp >> 20) & 63, C = (chp >> 26) & 31;
  if (xn < 0 || yn < 0 || xn >= cw || yn >= ch) return false;
  if (T > 1 && tile_row_of(ch >> 6, T, yn >> 6) != tile_row_of(ch >> 6, T, yc >> 6)) return false;
  if (C > 1 && tile_col_of(cw >> 6, C, xn >> 6) != tile_col_of(cw >> 6, C, xc >> 6)) return false;
  return zaddr64(xn, yn, cw >> 6) <= zaddr64(xc, yc, cw >> 6);
}

// ---------------------------------------------------------------------------------------------
// CABAC encoder (H.265 9.3.4, arithmetic encoder with a 32-bit low register and byte output)
// ---------------------------------------------------------------------------------------------
enum {
  CTX_SAO_MERGE = 0, CTX_SAO_TYPE = 1, CTX_SPLIT_CU = 2, CTX_TQ_BYPASS = 5, CTX_SKIP = 6, CTX_PRED_MODE = 9,
  CTX_PART_MODE = 10, CTX_PREV_INTRA = 14, CTX_CHROMA_MODE = 15, CTX_RQT_ROOT_CBF = 16, CTX_MERGE_FLAG = 17,
  CTX_MERGE_IDX = 18, CTX_INTER_PRED_IDC = 19, CTX_REF_IDX = 24, CTX_MVP_FLAG = 26, CTX_SPLIT_TRANSFORM = 27,
  CTX_CBF_LUMA = 30, CTX_CBF_CHROMA = 32, CTX_MVD_GT0 = 36, CTX_MVD_GT1 = 37, CTX_CU_QP_DELTA = 38, CTX_TS_FLAG = 40,
  CTX_LAST_X = 42, CTX_LAST_Y = 60, CTX_CSBF = 78, CTX_SIG = 82, CTX_GT1 = 124, CTX_GT2 = 148, CTX_COUNT = 154
};

#define KVZ_CNU 154
// H.265 Tables 9-5..9-37 initValue per initType (0: I, 1: P, 2: B), in CTX_* order
KVZ_CONST uint8_t kCabacInit[3][CTX_COUNT] = {
 {153, 200, 139,141,157, 154, KVZ_CNU,KVZ_CNU,KVZ_CNU, KVZ_CNU, 184,KVZ_CNU,KVZ_CNU,KVZ_CNU, 184, 63, KVZ_CNU, KVZ_CNU, KVZ_CNU,
  KVZ_CNU,KVZ_CNU,KVZ_CNU,KVZ_CNU,KVZ_CNU, KVZ_CNU,KVZ_CNU, KVZ_CNU, 153,138,138, 111,141, 94,138,182,154, KVZ_CNU,KVZ_CNU, 154,154, 139,139,
  110,110,124,125,140,153,125,127,140,109,111,143,127,111,79,108,123,63,
  110,110,124,125,140,153,125,127,140,109,111,143,127,111,79,108,123,63,
  91,171,134,141,
  111,111,125,110,110,94,124,108,124,107,125,141,179,153,125,107,125,141,179,153,125,107,125,141,179,153,125,
  140,139,182,182,152,136,152,136,153,136,139,111,136,139,111,
  140,92,137,138,140,152,138,139,153,74,149,92,139,107,122,152, 140,179,166,182,140,227,122,197,
  138,153,136,167,152,152},
 {153, 185, 107,139,126, 154, 197,185,201, 149, 154,139,154,154, 154, 152, 79, 110, 122,
  95,79,63,31,31, 153,153, 168, 124,138,94, 153,111, 149,107,167,154, 140,198, 154,154, 139,139,
  125,110,94,110,95,79,125,111,110,78,110,111,111,95,94,108,123,108,
  125,110,94,110,95,79,125,111,110,78,110,111,111,95,94,108,123,108,
  121,140,61,154,
  155,154,139,153,139,123,123,63,153,166,183,140,136,153,154,166,183,140,136,153,154,166,183,140,136,153,154,
  170,153,123,123,107,121,107,121,167,151,183,140,151,183,140,
  154,196,196,167,154,152,167,182,182,134,149,136,153,121,136,137, 169,194,166,167,154,167,137,182,
  107,167,91,122,107,167},
 {153, 160, 107,139,126, 154, 197,185,201, 134, 154,139,154,154, 183, 152, 79, 154, 137,
  95,79,63,31,31, 153,153, 168, 224,167,122, 153,111, 149,92,167,154, 169,198, 154,154, 139,139,
  125,110,124,110,95,94,125,111,111,79,125,126,111,111,79,108,123,93,
  125,110,124,110,95,94,125,111,111,79,125,126,111,111,79,108,123,93,
  121,140,61,154,
  170,154,139,153,139,123,123,63,124,166,183,140,136,153,154,166,183,140,136,153,154,166,183,140,136,153,154,
  170,153,138,138,122,121,122,121,167,151,183,140,151,183,140,
  154,196,167,167,154,152,167,182,182,134,149,136,153,121,136,122, 169,208,166,167,154,152,167,182,
  107,167,91,107,107,167}};

// context variable = (pStateIdx << 1) | valMps
KVZ_HD void cabac_init_contexts(uint8_t *ctx, int init_type, int qp)
{
  qp = clip3(0, 51, qp);
  for (int i = 0; i < CTX_COUNT; i++) {
    int v = kCabacInit[init_type][i];
    int slope = (v >> 4) * 5 - 45, offs = ((v & 15) << 3) - 16;
    int pre = clip3(1, 126, ((slope * qp) >> 4) + offs);
    int mps = pre <= 63 ? 0 : 1;
    ctx[i] = (uint8_t)(((mps ? pre - 64 : 63 - pre) << 1) | mps);
  }
}

// Small tables the serial coder touches for every bin, gathered so that the entropy kernel can
// keep one copy in LDS (a dependent global-memory lookup per bin is what made the first version
// latency bound); the host tests use a static copy.
struct CoreTabs {
  uint32_t lps4[64];                 // rangeTabLps row packed little-endian: byte q = range for quarter q
  uint8_t next_lps[64];
  uint8_t diag4x[16], diag4y[16], diag8x[64], diag8y[64], diag2x[4], diag2y[4];
  uint8_t ctxmap4x4[16];
  uint8_t pos4[3][16];               // scan position k of a 4x4 block -> x | y << 2 (= raster index), per scan_idx
  uint8_t sigpat[4][16];             // sig_coeff_flag context pattern (9.3.4.2.5) by prev_csbf and raster position
};
KVZ_HD constexpr void core_tabs_fill_entry(CoreTabs &t, int i)     // i in [0, 64): callers may spread i over lanes
{
  t.lps4[i] = (uint32_t)kRangeLps[i][0] | ((uint32_t)kRangeLps[i][1] << 8) | ((uint32_t)kRangeLps[i][2] << 16) | ((uint32_t)kRangeLps[i][3] << 24);
  t.next_lps[i] = kNextLps[i];
  t.diag8x[i] = kDiag8x[i]; t.diag8y[i] = kDiag8y[i];
  if (i < 16) { t.diag4x[i] = kDiag4x[i]; t.diag4y[i] = kDiag4y[i]; t.ctxmap4x4[i] = kCtxIdxMap4x4[i]; }
  if (i < 4) { t.diag2x[i] = kDiag2x[i]; t.diag2y[i] = kDiag2y[i]; }
  if (i < 16) {
    t.pos4[0][i] = (uint8_t)(kDiag4x[i] | (kDiag4y[i] << 2));
    t.pos4[1][i] = (uint8_t)i;                                   // horizontal scan: x = i & 3, y = i >> 2
    t.pos4[2][i] = (uint8_t)((i >> 2) | ((i & 3) << 2));         // vertical scan: y = i & 3, x = i >> 2
    const int xp = i & 3, yp = i >> 2;
    t.sigpat[0][i] = (uint8_t)((xp + yp == 0) ? 2 : (xp + yp < 3) ? 1 : 0);
    t.sigpat[1][i] = (uint8_t)((yp == 0) ? 2 : (yp == 1) ? 1 : 0);
    t.sigpat[2][i] = (uint8_t)((xp == 0) ? 2 : (xp == 1) ? 1 : 0);
    t.sigpat[3][i] = 2;
  }
}

struct CabacEnc {
  const CoreTabs *tabs;
  uint32_t low, range;
  int bits_left, num_buffered, buffered_byte;
  uint8_t *buf; int pos, cap;
  uint8_t *ctx;
  uint32_t nbins;
};

KVZ_HD void cabac_put_byte(CabacEnc &c, int b) { if (c.pos < c.cap) c.buf[c.pos] = (uint8_t)b; c.pos++; }

KVZ_HD void cabac_start(CabacEnc &c, uint8_t *buf, int cap, uint8_t *ctx, const CoreTabs *tabs)
{
  c.tabs = tabs;
  c.low = 0; c.range = 510; c.bits_left = 23; c.num_buffered = 0; c.buffered_byte = 0xff;
  c.buf = buf; c.pos = 0; c.cap = cap; c.ctx = ctx;
}

KVZ_HD void cabac_write_out(CabacEnc &c)
{
  uint32_t lead = c.low >> (24 - c.bits_left);
  c.bits_left += 8;
  c.low &= 0xffffffffu >> c.bits_left;
  if (lead == 0xff) { c.num_buffered++; return; }
  if (c.num_buffered > 0) {
    uint32_t carry = lead >> 8;
    cabac_put_byte(c, (int)(c.buffered_byte + carry));
    c.buffered_byte = (int)(lead & 0xff);
    int fill = (int)((0xff + carry) & 0xff);
    while (c.num_buffered > 1) { cabac_put_byte(c, fill); c.num_buffered--; }
  } else {
    c.num_buffered = 1; c.buffered_byte = (int)lead;
  }
}

KVZ_HD void cabac_bin(CabacEnc &c, int ci, int bin)
{
  uint8_t s = c.ctx[ci];
  int state = s >> 1, mps = s & 1;
  uint32_t lps = (c.tabs->lps4[state] >> (((c.range >> 6) & 3) * 8)) & 0xffu;
  c.nbins++;
  c.range -= lps;
  if (bin != mps) {
    int nb = 0; uint32_t t = lps;          // renormalisation shift: range becomes >= 256
    while (t < 256) { t <<= 1; nb++; }
    c.low = (c.low + c.range) << nb;
    c.range = t;
    if (state == 0) mps ^= 1;
    c.ctx[ci] = (uint8_t)((c.tabs->next_lps[state] << 1) | mps);
    c.bits_left -= nb;
  } else {
    c.ctx[ci] = (uint8_t)(((state < 62 ? state + 1 : state) << 1) | mps);
    if (c.range >= 256) return;
    c.low <<= 1; c.range <<= 1; c.bits_left--;
  }
  if (c.bits_left < 12) cabac_write_out(c);
}

KVZ_HD void cabac_bypass(CabacEnc &c, int bin)
{
  c.nbins++;
  c.low <<= 1;
  if (bin) c.low += c.range;
  c.bits_left--;
  if (c.bits_left < 12) cabac_write_out(c);
}

KVZ_HD void cabac_bypass_bits(CabacEnc &c, uint32_t val, int n)
{
  c.nbins += (uint32_t)n;
  while (n > 8) {
    n -= 8;
    uint32_t pat = val >> n;
    c.low = (c.low << 8) + c.range * pat;
    val -= pat << n;
    c.bits_left -= 8;
    if (c.bits_left < 12) cabac_write_out(c);
  }
  c.low = (c.low << n) + c.range * val;
  c.bits_left -= n;
  if (c.bits_left < 12) cabac_write_out(c);
}

KVZ_HD void cabac_terminate(CabacEnc &c, int bin)
{
  c.nbins++;
  c.range -= 2;
  if (bin) {
    c.low += c.range;
    c.low <<= 7; c.range = 2 << 7; c.bits_left -= 7;
  } else if (c.range >= 256) {
    return;
  } else {
    c.low <<= 1; c.range <<= 1; c.bits_left--;
  }
  if (c.bits_left < 12) cabac_write_out(c);
}

// Flush after a terminating bin equal to 1: remaining bits, then the stop bit '1' and zero bits
// up to the byte boundary (rbsp_slice_segment_trailing_bits / byte_alignment).
KVZ_HD void cabac_finish(CabacEnc &c)
{
  if (c.low >> (32 - c.bits_left)) {
    cabac_put_byte(c, c.buffered_byte + 1);
    while (c.num_buffered > 1) { cabac_put_byte(c, 0x00); c.num_buffered--; }
    c.low -= 1u << (32 - c.bits_left);
  } else {
    if (c.num_buffered > 0) cabac_put_byte(c, c.buffered_byte);
    while (c.num_buffered > 1) { cabac_put_byte(c, 0xff); c.num_buffered--; }
  }
  // remaining (24 - bits_left) bits of low >> 8, followed by '1' and alignment zeros
  int nbits = 24 - c.bits_left;
  uint32_t v = ((c.low >> 8) << 1) | 1u; nbits += 1;          // append the stop bit
  int pad = (8 - (nbits & 7)) & 7;
  v <<= pad; nbits += pad;
  for (int sh = nbits - 8; sh >= 0; sh -= 8) cabac_put_byte(c, (int)((v >> sh) & 0xff));
}

// ---------------------------------------------------------------------------------------------
// Token sink: the GPU does binarisation and context selection for every syntax element in
// parallel (k_tokenize) and emits 16-bit tokens in coding order; the strictly serial arithmetic
// coder then only walks the token list (entropy_host.h).  The syntax functions below are templates
// over the sink, so the same code drives a CabacEnc directly (host reference path / tests).
//   0b0ccccccccb              context-coded bin: ctx index c, value b
//   0b10nnnn vvvvvvvvvv       bypass string of n+1 (1..10) bits, value v, MSB first
//   0b11.............b        terminating bin b
// ---------------------------------------------------------------------------------------------
struct TokOut {
  const CoreTabs *tabs;
  uint16_t *p; int n, cap;
};
KVZ_HD void tok_push(TokOut &t, uint32_t v) { if (t.n < t.cap) t.p[t.n] = (uint16_t)v; t.n++; }
KVZ_HD void cabac_bin(TokOut &t, int ci, int bin) { tok_push(t, ((uint32_t)ci << 1) | (uint32_t)(bin ? 1 : 0)); }
KVZ_HD void cabac_bypass(TokOut &t, int bin) { tok_push(t, 0x8000u | (uint32_t)(bin ? 1 : 0)); }
KVZ_HD void cabac_bypass_bits(TokOut &t, uint32_t val, int n)
{
  while (n > 10) { n -= 10; tok_push(t, 0x8000u | (9u << 10) | ((val >> n) & 0x3ffu)); }
  if (n > 0) tok_push(t, 0x8000u | ((uint32_t)(n - 1) << 10) | (val & ((1u << n) - 1u)));
}
KVZ_HD void cabac_terminate(TokOut &t, int bin) { tok_push(t, 0xC000u | (uint32_t)(bin ? 1 : 0)); }
// a sink that only counts the tokens TokOut would receive (contexts and values are never formed: the
// compiler drops their computation from the counting instantiation of the emitters)
struct TokCount {
  const CoreTabs *tabs;
  int n;
};
template <class S> struct sink_counts_only { static constexpr bool value = false; };
template <> struct sink_counts_only<TokCount> { static constexpr bool value = true; };
KVZ_HD void cabac_bin(TokCount &t, int, int) { t.n++; }
KVZ_HD void cabac_bypass(TokCount &t, int) { t.n++; }
KVZ_HD void cabac_bypass_bits(TokCount &t, uint32_t, int n) { t.n += (n + 9) / 10; }
KVZ_HD void cabac_terminate(TokCount &t, int) { t.n++; }
// replay of a token list into the arithmetic coder
KVZ_HD void cabac_play_tokens(CabacEnc &c, const uint16_t *tok, int n)
{
  for (int i = 0; i < n; i++) {
    uint32_t t = tok[i];
    if (!(t & 0x8000u)) cabac_bin(c, (int)(t >> 1), (int)(t & 1));
    else if (!(t & 0x4000u)) cabac_bypass_bits(c, t & 0x3ffu, (int)((t >> 10) & 15) + 1);
    else cabac_terminate(c, (int)(t & 1));
  }
}

// ---------------------------------------------------------------------------------------------
// sao() (H.265 7.3.8.3): merge flags are pure syntax -- left when the left CTU has identical parameters, else up likewise
// (`left` / `up`: the neighbours that may be merged from, else NULL).  sao_offset_abs is TR with cMax 7, all bypass.
// ---------------------------------------------------------------------------------------------
KVZ_HD bool sao_same(const SaoParams &a, const SaoParams &b)
{
  bool eq = true;
  for (int c = 0; c < 3; c++) {
    eq = eq && a.type[c] == b.type[c] && a.eo_class[c] == b.eo_class[c] && a.band_pos[c] == b.band_pos[c];
    for (int k = 0; k < 4; k++) eq = eq && a.offset[c][k] == b.offset[c][k];
  }
  return eq;
}
template <class S>
KVZ_HD void enc_sao(S &t, const SaoParams &p, const SaoParams *left, const SaoParams *up)
{
  if (left) { const bool m = sao_same(p, *left); cabac_bin(t, CTX_SAO_MERGE, m); if (m) return; }
  if (up) { const bool m = sao_same(p, *up); cabac_bin(t, CTX_SAO_MERGE, m); if (m) return; }
  for (int ci = 0; ci < 3; ci++) {
    if (ci < 2) {
      cabac_bin(t, CTX_SAO_TYPE, p.type[ci] != 0);
      if (p.type[ci]) cabac_bypass(t, p.type[ci] == 2);
    }
    if (!p.type[ci]) continue;
    for (int i = 0; i < 4; i++) {
      const int a = p.offset[ci][i] < 0 ? -p.offset[ci][i] : p.offset[ci][i];
      if (a < 7) cabac_bypass_bits(t, ((1u << a) - 1u) << 1, a + 1); else cabac_bypass_bits(t, 127u, 7);
    }
    if (p.type[ci] == 1) {
      for (int i = 0; i < 4; i++) if (p.offset[ci][i]) cabac_bypass(t, p.offset[ci][i] < 0);
      cabac_bypass_bits(t, p.band_pos[ci], 5);
    } else if (ci < 2) cabac_bypass_bits(t, p.eo_class[ci], 2);
  }
}

// ---------------------------------------------------------------------------------------------
// residual_coding (H.265 7.3.8.11) for the encoder's tool set: diagonal / horizontal / vertical
// scans, no transform skip, no sign hiding.  `lv` points at the block's top-left level inside a
// plane-shaped level array with pitch `stride`.  The block has at least one non-zero level.
// ---------------------------------------------------------------------------------------------
KVZ_HD void scan_pos(const CoreTabs *t, int scan_idx, int log2blk, int i, int &x, int &y)
{
  // position i of the scan of a (1 << log2blk)^2 block (log2blk 0..3)
  int n = 1 << log2blk;
  if (scan_idx == 1) { x = i & (n - 1); y = i >> log2blk; return; }
  if (scan_idx == 2) { y = i & (n - 1); x = i >> log2blk; return; }
  if (log2blk == 2) { x = t->diag4x[i]; y = t->diag4y[i]; }
  else if (log2blk == 3) { x = t->diag8x[i]; y = t->diag8y[i]; }
  else if (log2blk == 1) { x = t->diag2x[i]; y = t->diag2y[i]; }
  else { x = 0; y = 0; }
}

template <class S>
KVZ_HD void enc_last_prefix(S &c, int base, int log2, int cidx, int prefix)
{
  int off, sh, mx = (log2 << 1) - 1;
  if (cidx == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); sh = (log2 + 1) >> 2; }
  else { off = 15; sh = log2 - 2; }
  for (int i = 0; i < prefix; i++) cabac_bin(c, base + off + (i >> sh), 1);
  if (prefix < mx) cabac_bin(c, base + off + (prefix >> sh), 0);
}

template <class S>
KVZ_HD void enc_abs_remaining(S &c, int v, int rice)
{
  int q = v >> rice;
  if (q < 4) {
    cabac_bypass_bits(c, (1u << (q + 1)) - 2u, q + 1);                 // q ones and a zero
    if (rice) cabac_bypass_bits(c, (uint32_t)(v & ((1 << rice) - 1)), rice);
  } else {
    int x = v - (4 << rice), k = rice + 1, ones = 4;
    while (x >= (1 << k)) { x -= 1 << k; k++; ones++; }
    cabac_bypass_bits(c, (1u << (ones + 1)) - 2u, ones + 1);
    cabac_bypass_bits(c, (uint32_t)x, k);
  }
}

// A transform block prepared for entropy coding: the levels of every 4x4 sub-block in SCAN order
// (scan[i * 16 + k] = level at scan position k of the sub-block at position i of the sub-block scan), a 16-bit
// significance mask per sub-block with bit k = scan position k, a per-sub-block coded flag
// addressed by (ys * 8 + xs), and the mask of non-empty sub-blocks (bit i).  On the GPU one wave
// builds it cooperatively in LDS (one lane per sub-block); digest_build_serial is the host twin.
struct TuDigest {
  int16_t scan[64 * 16];
  uint16_t mask[64];
  uint8_t csbf[64];
  uint64_t sbmask;
};

KVZ_HD int scan_raster4(const CoreTabs *t, int scan_idx, int k) { return t->pos4[scan_idx][k]; }

KVZ_HD void digest_build_serial(const CoreTabs *t, TuDigest &d, const int16_t *lv, int stride, int log2, int scan_idx)
{
  const int sbl = log2 - 2, nsb2 = 1 << (2 * sbl);
  d.sbmask = 0;
  for (int i = 0; i < 64; i++) d.csbf[i] = 0;
  for (int i = 0; i < nsb2; i++) {
    int xs, ys; scan_pos(t, scan_idx, sbl, i, xs, ys);
    uint32_t m = 0;
    for (int k = 0; k < 16; k++) {
      const int r = t->pos4[scan_idx][k];
      const int16_t v = lv[((ys << 2) + (r >> 2)) * stride + (xs << 2) + (r & 3)];
      d.scan[i * 16 + k] = v;
      if (v) m |= 1u << k;
    }
    d.mask[i] = (uint16_t)m;
    d.csbf[ys * 8 + xs] = m != 0;
    if (m) d.sbmask |= 1ull << i;
  }
}

// last_sig_coeff_{x,y}_{prefix,suffix}; returns the last sub-block / position through the references
template <class S>
KVZ_HD void enc_last_pos(S &c, const TuDigest &d, int log2, int cidx, int scan_idx, int &last_sb, int &last_pos)
{
  const int sbl = log2 - 2;
  const CoreTabs *t = c.tabs;
  last_sb = 63;
  while (!((d.sbmask >> last_sb) & 1)) last_sb--;
  last_pos = 15;
  { uint32_t m = d.mask[last_sb]; while (!((m >> last_pos) & 1)) last_pos--; }
  int xs0, ys0, xp0, yp0;
  scan_pos(t, scan_idx, sbl, last_sb, xs0, ys0); scan_pos(t, scan_idx, 2, last_pos, xp0, yp0);
  int lx = (xs0 << 2) + xp0, ly = (ys0 << 2) + yp0;
  if (scan_idx == 2) { int tt = lx; lx = ly; ly = tt; }
  int pfx[2], nbs[2], sfx[2];
  for (int dd = 0; dd < 2; dd++) {
    int v = dd ? ly : lx;
    if (v < 4) { pfx[dd] = v; nbs[dd] = 0; sfx[dd] = 0; }
    else { int len = ilog2((unsigned)v); pfx[dd] = 2 * len + ((v >> (len - 1)) & 1); nbs[dd] = len - 1; sfx[dd] = v & ((1 << (len - 1)) - 1); }
  }
  enc_last_prefix(c, CTX_LAST_X, log2, cidx, pfx[0]);
  enc_last_prefix(c, CTX_LAST_Y, log2, cidx, pfx[1]);
  if (pfx[0] > 3) cabac_bypass_bits(c, (uint32_t)sfx[0], nbs[0]);
  if (pfx[1] > 3) cabac_bypass_bits(c, (uint32_t)sfx[1], nbs[1]);
}

// does sub-block i contain, among its first 8 coefficients in coding order, one with |level| > 1?
// (that is exactly "greater1Ctx ended at 0", the only state the next sub-block inherits, 9.3.4.2.6)
KVZ_HD bool subblock_g1_any(const CoreTabs *, const TuDigest &d, int i, int)
{
  uint32_t mm = d.mask[i];
  for (int seen = 0; mm && seen < 8; seen++) {
    const int k = 31 - kv_clz32(mm); mm ^= 1u << k;
    const int v = d.scan[i * 16 + k];
    if (v > 1 || v < -1) return true;
  }
  return false;
}

// All syntax elements of sub-block i (coded_sub_block_flag .. coeff_abs_level_remaining).
// prev_g1: the previous non-empty sub-block in coding order ended with greater1Ctx == 0.
template <class S>
KVZ_HD void enc_subblock(S &c, const TuDigest &d, int i, int last_sb, int last_pos, bool prev_g1, int log2, int cidx, int scan_idx, int sign_hiding = 0)
{
  const int sbl = log2 - 2, nsb = 1 << sbl;
  const CoreTabs *t = c.tabs;
  int xs, ys; scan_pos(t, scan_idx, sbl, i, xs, ys);
  const int right = (xs < nsb - 1) ? d.csbf[ys * 8 + xs + 1] : 0;
  const int below = (ys < nsb - 1) ? d.csbf[(ys + 1) * 8 + xs] : 0;
  const uint32_t m = d.mask[i];
  int coded = m != 0, infer_dc = 0;
  if (i < last_sb && i > 0) {
    cabac_bin(c, CTX_CSBF + ((right | below) ? 1 : 0) + (cidx ? 2 : 0), coded);
    infer_dc = 1;
  } else coded = 1;                        // inferred 1 for the last and the DC sub-block
  if (!coded) return;
  // sig_coeff_flag: context = pattern by neighbouring coded flags and position + an offset per block kind (9.3.4.2.5)
  const uint8_t *pat = t->sigpat[right | (below << 1)], *pos = t->pos4[scan_idx];
  const int sigbase = CTX_SIG + (cidx ? 27 : 0);
  const int off = cidx == 0 ? ((i > 0 ? 3 : 0) + ((log2 == 3) ? ((scan_idx == 0) ? 9 : 15) : 21)) : ((log2 == 3) ? 9 : 12);
  const int start = (i == last_sb) ? last_pos - 1 : 15;
  if constexpr (sink_counts_only<S>::value) {
    // number of sig_coeff_flags: positions start .. 0, minus the inferred DC flag of a coded sub-block whose other
    // flags are all zero
    if (start >= 0) c.n += start + 1 - ((infer_dc && (m >> 1) == 0) ? 1 : 0);
  } else {
    for (int k = start; k >= 0; k--) {
      if (k > 0 || !infer_dc) {
        const int p = pos[k];
        int sc;
        if (log2 == 2) sc = t->ctxmap4x4[p];
        else if (i == 0 && k == 0) sc = 0;                           // the DC coefficient of the block
        else sc = pat[p] + off;
        const int sig = (m >> k) & 1;
        cabac_bin(c, sigbase + sc, sig);
        if (sig) infer_dc = 0;
      }
    }
  }
  if (!m) return;
  int ctx_set = (i > 0 && cidx == 0) ? 2 : 0;
  if (prev_g1) ctx_set++;
  // greater1 flags of the first eight significant coefficients in coding order (scan position 15 .. 0); signs of all
  const int16_t *lv = &d.scan[i * 16];
  int c1 = 1, nsig = 0, g1idx = -1, g2 = 0;
  uint32_t signs = 0;
  for (uint32_t mm = m; mm; nsig++) {
    const int k = 31 - kv_clz32(mm); mm ^= 1u << k;
    const int v = lv[k], a = v < 0 ? -v : v;
    signs = (signs << 1) | (v < 0 ? 1u : 0u);
    if (nsig < 8) {
      const int g1 = a > 1;
      cabac_bin(c, CTX_GT1 + (cidx ? 16 : 0) + ctx_set * 4 + c1, g1);
      if (g1) { c1 = 0; if (g1idx < 0) { g1idx = nsig; g2 = a > 2; } }
      else if (c1 > 0 && c1 < 3) c1++;
    }
  }
  if (g1idx >= 0) cabac_bin(c, CTX_GT2 + (cidx ? 4 : 0) + ctx_set, g2);
  // sign_data_hiding (7.3.8.11): the sign of the group's first coefficient in scan order -- the last one collected above -- is not sent
  // when it lies more than three positions below the group's last one
  if (sign_hiding && (31 - kv_clz32(m)) - kv_ctz32(m) > 3) cabac_bypass_bits(c, signs >> 1, nsig - 1);
  else cabac_bypass_bits(c, signs, nsig);
  int rice = 0, j = 0;
  for (uint32_t mm = m; mm; j++) {
    const int k = 31 - kv_clz32(mm); mm ^= 1u << k;
    const int v = lv[k], a = v < 0 ? -v : v;
    const int base = (j < 8) ? ((j == g1idx) ? 3 : 2) : 1;
    if (a >= base) {
      enc_abs_remaining(c, a - base, rice);
      if (a > 3 * (1 << rice)) rice = imin(rice + 1, 4);
    }
  }
}

// cu_qp_delta_abs (prefix: truncated unary cMax 5, first bin context 0, the others 1; suffix EG0) and cu_qp_delta_sign_flag
template <class S>
KVZ_HD void enc_cu_qp_delta(S &c, int d)
{
  const int a = d < 0 ? -d : d;
  int v = 0;
  while (v < 5 && v < a) { cabac_bin(c, CTX_CU_QP_DELTA + (v ? 1 : 0), 1); v++; }
  if (a < 5) cabac_bin(c, CTX_CU_QP_DELTA + (a ? 1 : 0), 0);
  else { int x = a - 5, k = 0; while (x >= (1 << k)) { cabac_bypass(c, 1); x -= 1 << k; k++; } cabac_bypass(c, 0); cabac_bypass_bits(c, (uint32_t)x, k); }
  if (a) cabac_bypass(c, d < 0);
}

template <class S>
KVZ_HD void enc_residual_digest(S &c, const TuDigest &d, int log2, int cidx, int scan_idx, int sign_hiding = 0)
{
  int last_sb, last_pos;
  enc_last_pos(c, d, log2, cidx, scan_idx, last_sb, last_pos);
  bool prev_g1 = false;
  for (int i = last_sb; i >= 0; i--) {
    enc_subblock(c, d, i, last_sb, last_pos, prev_g1, log2, cidx, scan_idx, sign_hiding);
    if (d.mask[i]) prev_g1 = subblock_g1_any(c.tabs, d, i, scan_idx);
  }
}

template <class S>
KVZ_HD void enc_residual(S &c, const int16_t *lv, int stride, int log2, int cidx, int scan_idx, int sign_hiding = 0)
{
  TuDigest d;
  digest_build_serial(c.tabs, d, lv, stride, log2, scan_idx);
  enc_residual_digest(c, d, log2, cidx, scan_idx, sign_hiding);
}

KVZ_HD int intra_scan_idx(int intra, int log2, int cidx, int mode)
{
  if (!intra) return 0;
  if (log2 == 2 || (log2 == 3 && cidx == 0)) {
    if (mode >= 6 && mode <= 14) return 2;
    if (mode >= 22 && mode <= 30) return 1;
  }
  return 0;
}

template <class S>
KVZ_HD void enc_mvd(S &c, int dx, int dy)
{
  int ax = iabs(dx), ay = iabs(dy);
  cabac_bin(c, CTX_MVD_GT0, ax > 0); cabac_bin(c, CTX_MVD_GT0, ay > 0);
  if (ax > 0) cabac_bin(c, CTX_MVD_GT1, ax > 1);
  if (ay > 0) cabac_bin(c, CTX_MVD_GT1, ay > 1);
  for (int d = 0; d < 2; d++) {
    int a = d ? ay : ax, neg = (d ? dy : dx) < 0;
    if (!a) continue;
    if (a > 1) {                              // abs_mvd_minus2: EG1
      int x = a - 2, k = 1, ones = 0;
      while (x >= (1 << k)) { x -= 1 << k; k++; ones++; }
      cabac_bypass_bits(c, (1u << (ones + 1)) - 2u, ones + 1);
      cabac_bypass_bits(c, (uint32_t)x, k);
    }
    cabac_bypass(c, neg);
  }
}

template <class S>
KVZ_HD void enc_merge_idx(S &c, int idx)
{
  cabac_bin(c, CTX_MERGE_IDX, idx > 0);
  for (int i = 1; i < 4 && idx >= i; i++) cabac_bypass(c, idx > i);
}

// Per-CU record as the entropy coder sees it.  A "view" type V provides `CuRec at(int x, int y)`
// for any luma position of the current CTU and its left / above neighbours: FrameView reads the
// per-8x8 arrays directly (host tests), the entropy kernel uses a tile staged in LDS.
struct CuRec {
  uint8_t log2, intra, flags, merge_idx, mvp_idx, intra_mode, cbf, pad;
  int16_t mvdx, mvdy;
};
struct FrameView {
  const EncFrame *f;
  KVZ_HD CuRec at(int x, int y) const
  {
    int i = b8idx(*f, x, y);
    CuRec r;
    r.log2 = f->cu_log2[i]; r.intra = f->cu_intra[i]; r.flags = f->cu_flags[i]; r.merge_idx = f->cu_merge_idx[i];
    r.mvp_idx = f->cu_mvp_idx[i]; r.intra_mode = f->cu_intra_mode[i]; r.cbf = f->cu_cbf[i]; r.pad = 0;
    r.mvdx = f->cu_mvd[i * 2]; r.mvdy = f->cu_mvd[i * 2 + 1];
    return r;
  }
};

// Intra MPM candidates (H.265 8.4.2) for the CU at (x0, y0)
template <class V>
KVZ_HD void intra_mpm(const V &v, int cw, int ch, int x0, int y0, int cand[3])
{
  int ca = 1, cb = 1;
  if (avail64(cw, ch, x0, y0, x0 - 1, y0)) { CuRec n = v.at(x0 - 1, y0); if (n.intra) ca = n.intra_mode; }
  if (avail64(cw, ch, x0, y0, x0, y0 - 1) && (y0 - 1) >= ((y0 >> 6) << 6)) { CuRec n = v.at(x0, y0 - 1); if (n.intra) cb = n.intra_mode; }
  if (ca == cb) {
    if (ca < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
    else { cand[0] = ca; cand[1] = 2 + ((ca + 29) % 32); cand[2] = 2 + ((ca - 2 + 1) % 32); }
  } else {
    cand[0] = ca; cand[1] = cb;
    cand[2] = (ca != 0 && cb != 0) ? 0 : ((ca != 1 && cb != 1) ? 1 : 26);
  }
}

// split_cu_flag bins (7.3.8.4) for every quadtree level whose block starts at z-order index z
// (in 8x8 units) of the CTU at (cx, cy), down to the CU of size (1 << cl) that starts there.
template <class V, class S>
KVZ_HD void enc_split_flags(const V &v, S &c, int cw, int ch, int x0, int y0, int z, int cl)
{
  for (int l2 = 6; l2 > 3; l2--) {
    int zmask = (1 << (2 * (l2 - 3))) - 1;
    if (z & zmask) continue;
    int depth = 6 - l2;
    int l = avail64(cw, ch, x0, y0, x0 - 1, y0) && (6 - v.at(x0 - 1, y0).log2) > depth;
    int a = avail64(cw, ch, x0, y0, x0, y0 - 1) && (6 - v.at(x0, y0 - 1).log2) > depth;
    cabac_bin(c, CTX_SPLIT_CU + l + a, cl < l2);
    if (cl >= l2) break;
  }
}

// coding_unit() up to and including the cbf flags of its single transform unit (7.3.8.5-7.3.8.10).
// Returns the cbf bits (bit0 Y, bit1 Cb, bit2 Cr) whose residual_coding() must follow, 0 if none.
template <class V, class S>
KVZ_HD int enc_cu_header(const V &v, S &c, int cw, int ch, bool pic_intra, int x0, int y0, const CuRec &cu, bool bypass = false)
{
  const int intra = cu.intra, flags = cu.flags, cbf = cu.cbf, log2 = cu.log2;
  if (bypass) cabac_bin(c, CTX_TQ_BYPASS, 1);                  // cu_transquant_bypass_flag (`lossless`: the PPS enables it and every coding unit sets it), first in the coding unit
  if (!pic_intra) {
    int l = avail64(cw, ch, x0, y0, x0 - 1, y0) && (v.at(x0 - 1, y0).flags & CU_SKIP);
    int a = avail64(cw, ch, x0, y0, x0, y0 - 1) && (v.at(x0, y0 - 1).flags & CU_SKIP);
    cabac_bin(c, CTX_SKIP + l + a, flags & CU_SKIP);
    if (flags & CU_SKIP) { enc_merge_idx(c, cu.merge_idx); return 0; }
    cabac_bin(c, CTX_PRED_MODE, intra);
  }
  if (!intra || log2 == 3) cabac_bin(c, CTX_PART_MODE, 1);        // PART_2Nx2N
  if (intra) {
    int mode = cu.intra_mode;
    int cand[3]; intra_mpm(v, cw, ch, x0, y0, cand);
    int mpm = -1;
    for (int k = 0; k < 3; k++) if (cand[k] == mode) { mpm = k; break; }
    cabac_bin(c, CTX_PREV_INTRA, mpm >= 0);
    if (mpm >= 0) { cabac_bypass(c, mpm > 0); if (mpm > 0) cabac_bypass(c, mpm > 1); }
    else {
      int t;
      if (cand[0] > cand[1]) { t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
      if (cand[0] > cand[2]) { t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
      if (cand[1] > cand[2]) { t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
      int rem = mode;
      for (int k = 2; k >= 0; k--) if (rem > cand[k]) rem--;
      cabac_bypass_bits(c, (uint32_t)rem, 5);
    }
    cabac_bin(c, CTX_CHROMA_MODE, 0);                               // intra_chroma_pred_mode = 4
  } else {
    cabac_bin(c, CTX_MERGE_FLAG, (flags & CU_MERGE) ? 1 : 0);
    if (flags & CU_MERGE) enc_merge_idx(c, cu.merge_idx);
    else {
      enc_mvd(c, cu.mvdx, cu.mvdy);
      cabac_bin(c, CTX_MVP_FLAG, cu.mvp_idx);
      cabac_bin(c, CTX_RQT_ROOT_CBF, cbf != 0);
    }
    if (!cbf) return 0;
  }
  cabac_bin(c, CTX_CBF_CHROMA, (cbf >> 1) & 1);
  cabac_bin(c, CTX_CBF_CHROMA, (cbf >> 2) & 1);
  if (intra || (cbf & 6)) cabac_bin(c, CTX_CBF_LUMA + 1, cbf & 1);
  return cbf;
}

KVZ_HD void ctu_z_to_xy(int z, int &xi, int &yi)
{
  xi = 0; yi = 0;
  for (int b = 0; b < 3; b++) { xi |= ((z >> (2 * b)) & 1) << b; yi |= ((z >> (2 * b + 1)) & 1) << b; }
}

// coding_quadtree() of one 64x64 CTU (serial reference form used by the host tests; the entropy
// kernel walks the same way but stages data in LDS with the whole wave)
KVZ_HD void enc_ctu(const EncFrame &f, CabacEnc &c, int cx, int cy)
{
  FrameView v; v.f = &f;
  for (int z = 0; z < 64;) {
    int xi, yi; ctu_z_to_xy(z, xi, yi);
    int x0 = cx + xi * 8, y0 = cy + yi * 8;
    CuRec cu = v.at(x0, y0);
    enc_split_flags(v, c, f.cw, f.chp, x0, y0, z, cu.log2);
    int cbf = enc_cu_header(v, c, f.cw, f.chp, f.is_intra != 0, x0, y0, cu, f.lossless != 0);
    if (cbf & 1) enc_residual(c, f.coef[0] + y0 * f.cw + x0, f.cw, cu.log2, 0, intra_scan_idx(cu.intra, cu.log2, 0, cu.intra_mode), f.signhide);
    for (int ci = 1; ci <= 2; ci++)
      if ((cbf >> ci) & 1)
        enc_residual(c, f.coef[ci] + (y0 >> 1) * (f.cw >> 1) + (x0 >> 1), f.cw >> 1, cu.log2 - 1, ci, intra_scan_idx(cu.intra, cu.log2 - 1, ci, cu.intra_mode), f.signhide);
    z += 1 << (2 * (cu.log2 - 3));
  }
}

// ---------------------------------------------------------------------------------------------
// Merge / AMVP signalling of one inter 2Nx2N CU from the final motion field of a P picture with
// one reference picture and no intra CUs... (intra neighbours are treated as unavailable).
// H.265 8.5.3.2.2-8.5.3.2.7 specialised: Log2ParMrgLevel = 2, MaxNumMergeCand = 5, no TMVP.
// ---------------------------------------------------------------------------------------------
KVZ_HD int mvd_bits(int q)
{
  int a = iabs(q);
  if (a == 0) return 1;
  if (a == 1) return 3;
  int x = a - 2, k = 1, len = 0;
  while (x >= (1 << k)) { x -= 1 << k; k++; len++; }
  return 2 + len + 1 + k + 1;
}

struct NbMv { bool ok; int mx, my; };
// where the derivations below read a CU's record from: the frame's arrays (host tests, band encoder) or a tile of them staged in LDS (k_inter_signal)
struct MvRec { int intra, mx, my, cbf; };
struct FrameMvView {
  const EncFrame &f;
  KVZ_HD MvRec at(int x, int y) const { const int i = b8idx(f, x, y); MvRec r; r.intra = f.cu_intra[i]; r.mx = f.cu_mv[i * 2]; r.my = f.cu_mv[i * 2 + 1]; r.cbf = f.cu_cbf[i]; return r; }
};
template <class V>
KVZ_HD NbMv nb_mv(const V &v, int cw, int chp, int xc, int yc, int xn, int yn)
{
  // Straight-line on purpose: the record is loaded whether or not the neighbour exists (the CU's own one stands in), so the loads of
  // all five neighbours of a CU are in flight together instead of ten dependent round trips (k_inter_signal: 4K 24 -> 9 us).
  const bool av = avail64(cw, chp, xc, yc, xn, yn);
  const MvRec m = v.at(av ? xn : xc, av ? yn : yc);
  NbMv r; r.ok = av && !m.intra; r.mx = r.ok ? m.mx : 0; r.my = r.ok ? m.my : 0;
  return r;
}
KVZ_HD bool same_mv(const NbMv &a, const NbMv &b) { return a.mx == b.mx && a.my == b.my; }

// the five spatial neighbours of the 2Nx2N PU at (x0, y0), size n, that both the merge and the AMVP derivation read (8.5.3.2.3, 8.5.3.2.7)
struct FiveNb { NbMv A0, A1, B0, B1, B2; };
template <class V>
KVZ_HD FiveNb five_neighbours(const V &v, int cw, int chp, int x0, int y0, int n)
{
  FiveNb q;
  q.A1 = nb_mv(v, cw, chp, x0, y0, x0 - 1, y0 + n - 1); q.B1 = nb_mv(v, cw, chp, x0, y0, x0 + n - 1, y0 - 1);
  q.B0 = nb_mv(v, cw, chp, x0, y0, x0 + n, y0 - 1); q.A0 = nb_mv(v, cw, chp, x0, y0, x0 - 1, y0 + n); q.B2 = nb_mv(v, cw, chp, x0, y0, x0 - 1, y0 - 1);
  return q;
}
// the five merge candidates (8.5.3.2.2-8.5.3.2.5)
KVZ_HD void merge_cand_list(const FiveNb &q, int cmx[5], int cmy[5])
{
  const NbMv &A1 = q.A1, &B1 = q.B1, &B0 = q.B0, &A0 = q.A0, &B2 = q.B2;
  bool fA1 = A1.ok;
  bool fB1 = B1.ok && !(A1.ok && same_mv(A1, B1));
  bool fB0 = B0.ok && !(B1.ok && same_mv(B1, B0));
  bool fA0 = A0.ok && !(A1.ok && same_mv(A1, A0));
  bool fB2 = B2.ok && !(A1.ok && same_mv(A1, B2)) && !(B1.ok && same_mv(B1, B2)) && !(fA0 && fA1 && fB0 && fB1);
  int nc = 0;
  if (fA1) { cmx[nc] = A1.mx; cmy[nc] = A1.my; nc++; }
  if (fB1) { cmx[nc] = B1.mx; cmy[nc] = B1.my; nc++; }
  if (fB0) { cmx[nc] = B0.mx; cmy[nc] = B0.my; nc++; }
  if (fA0) { cmx[nc] = A0.mx; cmy[nc] = A0.my; nc++; }
  if (fB2 && nc < 5) { cmx[nc] = B2.mx; cmy[nc] = B2.my; nc++; }
  while (nc < 5) { cmx[nc] = 0; cmy[nc] = 0; nc++; }        // zero candidates (refIdx 0 for one reference)
}
KVZ_HD void merge_cand_list(const EncFrame &f, int x0, int y0, int n, int cmx[5], int cmy[5]) { FrameMvView v{f}; merge_cand_list(five_neighbours(v, f.cw, f.chp, x0, y0, n), cmx, cmy); }
// the two AMVP candidates (8.5.3.2.6-8.5.3.2.7); every neighbour refers to the same picture
KVZ_HD void amvp_cand_list(const FiveNb &q, int px[2], int py[2])
{
  const NbMv &A1 = q.A1, &B1 = q.B1, &B0 = q.B0, &A0 = q.A0, &B2 = q.B2;
  bool haveA = A0.ok || A1.ok, haveB = B0.ok || B1.ok || B2.ok;
  NbMv a = A0.ok ? A0 : A1, b = B0.ok ? B0 : (B1.ok ? B1 : B2);
  if (!haveA && haveB) { a = b; haveA = true; }          // isScaledFlag == 0: A takes B's vector
  int np = 0;
  if (haveA) { px[np] = a.mx; py[np] = a.my; np++; }
  if (haveB && !(haveA && a.mx == b.mx && a.my == b.my)) { px[np] = b.mx; py[np] = b.my; np++; }
  while (np < 2) { px[np] = 0; py[np] = 0; np++; }
}
KVZ_HD void amvp_cand_list(const EncFrame &f, int x0, int y0, int n, int px[2], int py[2]) { FrameMvView v{f}; amvp_cand_list(five_neighbours(v, f.cw, f.chp, x0, y0, n), px, py); }

// the signalling of the inter CU at (x0, y0): merge (+ skip) with the first candidate that equals its vector, else AMVP with the cheaper predictor
struct CuSignal { int flags, midx, mvp, mvdx, mvdy; };
template <class V>
KVZ_HD CuSignal decide_signalling_values(const V &v, int cw, int chp, int x0, int y0, int log2)
{
  const int n = 1 << log2;
  const MvRec own = v.at(x0, y0);
  const int mvx = own.mx, mvy = own.my;
  const FiveNb q = five_neighbours(v, cw, chp, x0, y0, n);      // (once for both derivations: ten availability tests and record fetches were most of k_inter_signal's code)
  int cmx[5], cmy[5];
  merge_cand_list(q, cmx, cmy);
  CuSignal r; r.flags = 0; r.midx = 0; r.mvp = 0; r.mvdx = 0; r.mvdy = 0;
  for (int k = 4; k >= 0; k--) if (cmx[k] == mvx && cmy[k] == mvy) { r.flags = CU_MERGE; r.midx = k; }      // (the first match wins)
  if (r.flags && own.cbf == 0) r.flags |= CU_SKIP;
  if (!r.flags) {
    int px[2], py[2];
    amvp_cand_list(q, px, py);
    int b0 = mvd_bits(mvx - px[0]) + mvd_bits(mvy - py[0]);
    int b1 = mvd_bits(mvx - px[1]) + mvd_bits(mvy - py[1]);
    r.mvp = b1 < b0;
    r.mvdx = mvx - px[r.mvp]; r.mvdy = mvy - py[r.mvp];
  }
  return r;
}
KVZ_HD CuSignal decide_signalling_values(const EncFrame &f, int x0, int y0, int log2) { FrameMvView v{f}; return decide_signalling_values(v, f.cw, f.chp, x0, y0, log2); }
KVZ_HD void decide_signalling(const EncFrame &f, int x0, int y0, int log2)
{
  const int n = 1 << log2;
  const CuSignal r = decide_signalling_values(f, x0, y0, log2);
  const int flags = r.flags, midx = r.midx, mvp = r.mvp, mvdx = r.mvdx, mvdy = r.mvdy;
  for (int y = y0; y < y0 + n; y += 8)
    for (int x = x0; x < x0 + n; x += 8) {
      int i = b8idx(f, x, y);
      f.cu_flags[i] = (uint8_t)flags; f.cu_merge_idx[i] = (uint8_t)midx; f.cu_mvp_idx[i] = (uint8_t)mvp;
      f.cu_mvd[i * 2] = (int16_t)mvdx; f.cu_mvd[i * 2 + 1] = (int16_t)mvdy;
    }
}

// ---------------------------------------------------------------------------------------------
// Intra sample prediction (H.265 8.4.4.2).  `left[0]` = `top[0]` = p[-1][-1], left[1+i] = p[-1][i],
// Which intra prediction modes read the block's ABOVE-RIGHT reference samples p[x][-1], x >= n, or its BELOW-LEFT ones p[-1][y], y >= n -- directly or through
// the [1 2 1] reference filter (8.4.4.2.3-6); bit m = mode m does.  Found by perturbing the reference samples of the checker's predictor and confirmed the
// same way for this file's and the Python decoder's (tests/test_intra_dependencies.py).  Luma 16x16 / 32x32 filter more modes than the smaller blocks.
// Used twice: the intra chains wait for a neighbouring CTU only as far as the block's mode reads it, and the encoder keeps the blocks whose above-right /
// below-left samples lie in ANOTHER CTU to the modes that do not read them ("intra-chain") -- the CTU wavefront's lags shrink.
// (luma 32x32: every filtered mode -- all but DC, 10 and 26 -- may read both far corners, p[63][-1] and p[-1][63], through the strong filter)
KVZ_HD uint64_t intra_uses_above_right(int log2n, int cidx) { return cidx == 0 && log2n == 4 ? 0x7f9f80001ull : (cidx == 0 && log2n == 5 ? 0x7fbfffbfdull : 0x7f8000001ull); }
KVZ_HD uint64_t intra_uses_below_left(int log2n, int cidx) { return cidx == 0 && log2n == 4 ? 0x3f3fdull : (cidx == 0 && log2n == 5 ? 0x7fbfffbfdull : 0x3fdull); }

// top[1+i] = p[i][-1], i < 2n.  The arrays passed to intra_pred_sample() are the ones selected by
// the filtering decision of 8.4.4.2.3.
// ---------------------------------------------------------------------------------------------
KVZ_HD bool intra_filter_needed(int n, int cidx, int mode)
{
  if (cidx != 0 || mode == 1 || n == 4) return false;
  int d = imin(iabs(mode - 26), iabs(mode - 10));
  int thr = (n == 8) ? 7 : (n == 16) ? 1 : 0;
  return d > thr;
}
KVZ_HD bool intra_strong_filter(const uint8_t *left, const uint8_t *top, int n)
{
  if (n != 32) return false;
  int c = left[0];
  return iabs(c + top[2 * n] - 2 * top[n]) < 8 && iabs(c + left[2 * n] - 2 * left[n]) < 8;
}
// filtered value of reference index i (0..2n) of `a`, where b is the other array (for the corner)
KVZ_HD int intra_filtered_ref(const uint8_t *a, const uint8_t *b, int n, int i, bool strong)
{
  if (strong) {
    if (i == 0 || i == 64) return a[i];
    return ((64 - i) * a[0] + i * a[64] + 32) >> 6;
  }
  if (i == 0) return (a[1] + 2 * a[0] + b[1] + 2) >> 2;
  if (i == 2 * n) return a[i];
  return (a[i + 1] + 2 * a[i] + a[i - 1] + 2) >> 2;
}

KVZ_HD int intra_ref_main(const uint8_t *main, const uint8_t *side, int inv, int i)
{
  // ref[i] for the angular modes: i >= 0 -> main[i]; i < 0 -> projected from the side array
  return i >= 0 ? main[i] : side[(i * inv + 128) >> 8];
}

KVZ_HD int intra_pred_sample(const uint8_t *left, const uint8_t *top, int n, int log2n, int cidx, int mode, int dc, int x, int y)
{
  if (mode == 0) {
    return ((n - 1 - x) * left[1 + y] + (x + 1) * top[1 + n] + (n - 1 - y) * top[1 + x] + (y + 1) * left[1 + n] + n) >> (log2n + 1);
  }
  if (mode == 1) {
    if (cidx == 0 && n < 32) {
      if (x == 0 && y == 0) return (left[1] + 2 * dc + top[1] + 2) >> 2;
      if (y == 0) return (top[1 + x] + 3 * dc + 2) >> 2;
      if (x == 0) return (left[1 + y] + 3 * dc + 2) >> 2;
    }
    return dc;
  }
  const int angle = kIntraAngle[mode], inv = kInvAngle[mode];
  if (mode >= 18) {
    if (mode == 26 && cidx == 0 && n < 32 && x == 0) return clip8(top[1] + ((left[1 + y] - left[0]) >> 1));
    int idx = ((y + 1) * angle) >> 5, fact = ((y + 1) * angle) & 31;
    int r0 = intra_ref_main(top, left, inv, x + idx + 1);
    if (!fact) return r0;
    int r1 = intra_ref_main(top, left, inv, x + idx + 2);
    return ((32 - fact) * r0 + fact * r1 + 16) >> 5;
  } else {
    if (mode == 10 && cidx == 0 && n < 32 && y == 0) return clip8(left[1] + ((top[1 + x] - left[0]) >> 1));
    int idx = ((x + 1) * angle) >> 5, fact = ((x + 1) * angle) & 31;
    int r0 = intra_ref_main(left, top, inv, y + idx + 1);
    if (!fact) return r0;
    int r1 = intra_ref_main(left, top, inv, y + idx + 2);
    return ((32 - fact) * r0 + fact * r1 + 16) >> 5;
  }
}

// reference sample i of the 4n+1 samples in the order of 8.4.4.2.2 (i = 0: p[-1][2n-1] ...
// i = 2n: corner ... i = 4n: p[2n-1][-1]) -> component-sample coordinates
KVZ_HD void intra_ref_coord(int x0, int y0, int n, int i, int &x, int &y)
{
  if (i < 2 * n) { x = x0 - 1; y = y0 + 2 * n - 1 - i; }
  else if (i == 2 * n) { x = x0 - 1; y = y0 - 1; }
  else { x = x0 + (i - 2 * n - 1); y = y0 - 1; }
}

// ---------------------------------------------------------------------------------------------
// Deblocking (H.265 8.7.2): one 4-line luma edge segment / the chroma lines below it.
// q0 points at sample q0 of line 0, xs steps across the edge, ls along it.
// ---------------------------------------------------------------------------------------------
// beta_off / tc_off: slice_beta_offset_div2 * 2, slice_tc_offset_div2 * 2 (8.7.2.5.3)
KVZ_HD void deblock_luma_segment(uint8_t *q0p, int xs, int ls, int bs, int qp, int beta_off = 0, int tc_off = 0)
{
  int beta = kBetaTable[clip3(0, 51, qp + beta_off)];
  int tc = kTcTable[clip3(0, 53, qp + 2 * (bs - 1) + tc_off)];
#define P_(i, l) q0p[-((i) + 1) * xs + (l) * ls]
#define Q_(i, l) q0p[(i) * xs + (l) * ls]
  int dp0 = iabs(P_(2, 0) - 2 * P_(1, 0) + P_(0, 0)), dp3 = iabs(P_(2, 3) - 2 * P_(1, 3) + P_(0, 3));
  int dq0 = iabs(Q_(2, 0) - 2 * Q_(1, 0) + Q_(0, 0)), dq3 = iabs(Q_(2, 3) - 2 * Q_(1, 3) + Q_(0, 3));
  int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
  if (dpq0 + dpq3 >= beta) return;
  bool s0 = (2 * dpq0 < (beta >> 2)) && (iabs(P_(3, 0) - P_(0, 0)) + iabs(Q_(0, 0) - Q_(3, 0)) < (beta >> 3)) && (iabs(P_(0, 0) - Q_(0, 0)) < ((5 * tc + 1) >> 1));
  bool s3 = (2 * dpq3 < (beta >> 2)) && (iabs(P_(3, 3) - P_(0, 3)) + iabs(Q_(0, 3) - Q_(3, 3)) < (beta >> 3)) && (iabs(P_(0, 3) - Q_(0, 3)) < ((5 * tc + 1) >> 1));
  bool strong = s0 && s3;
  bool dep = dp < ((beta + (beta >> 1)) >> 3), deq = dq < ((beta + (beta >> 1)) >> 3);
  for (int l = 0; l < 4; l++) {
    int p0 = P_(0, l), p1 = P_(1, l), p2 = P_(2, l), p3 = P_(3, l), q0 = Q_(0, l), q1 = Q_(1, l), q2 = Q_(2, l), q3 = Q_(3, l);
    if (strong) {
      P_(0, l) = (uint8_t)clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
      P_(1, l) = (uint8_t)clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
      P_(2, l) = (uint8_t)clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      Q_(0, l) = (uint8_t)clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
      Q_(1, l) = (uint8_t)clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
      Q_(2, l) = (uint8_t)clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (iabs(delta) < tc * 10) {
        delta = clip3(-tc, tc, delta);
        P_(0, l) = (uint8_t)clip8(p0 + delta);
        Q_(0, l) = (uint8_t)clip8(q0 - delta);
        if (dep) P_(1, l) = (uint8_t)clip8(p1 + clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
        if (deq) Q_(1, l) = (uint8_t)clip8(q1 + clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
      }
    }
  }
#undef P_
#undef Q_
}

// c_off: pps_cb_qp_offset / pps_cr_qp_offset (cQpPicOffset of 8.7.2.5.5)
KVZ_HD void deblock_chroma_segment(uint8_t *q0p, int xs, int ls, int nlines, int qp_luma, int c_off = 0, int tc_off = 0)
{
  int qpc = kChromaQp[clip3(0, 57, qp_luma + c_off)];
  int tc = kTcTable[clip3(0, 53, qpc + 2 + tc_off)];
  for (int l = 0; l < nlines; l++) {
    uint8_t *q = q0p + l * ls;
    int p0 = q[-xs], p1 = q[-2 * xs], q0 = q[0], q1 = q[xs];
    int delta = clip3(-tc, tc, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
    q[-xs] = (uint8_t)clip8(p0 + delta);
    q[0] = (uint8_t)clip8(q0 - delta);
  }
}

// Boundary strength (H.265 8.7.2.4) of the edge between the 8x8 blocks containing samples
// (xp, yp) and (xq, yq); the caller has established that it is a CU (= TU = PU) boundary.
KVZ_HD int edge_bs(const EncFrame &f, int xp, int yp, int xq, int yq)
{
  int ip = b8idx(f, xp, yp), iq = b8idx(f, xq, yq);
  if (f.cu_intra[ip] || f.cu_intra[iq]) return 2;
  if ((f.cu_cbf[ip] & 1) || (f.cu_cbf[iq] & 1)) return 1;
  if (iabs(f.cu_mv[ip * 2] - f.cu_mv[iq * 2]) >= 4 || iabs(f.cu_mv[ip * 2 + 1] - f.cu_mv[iq * 2 + 1]) >= 4) return 1;
  return 0;
}
// is luma column x (multiple of 8) a CU boundary at row y?  (CUs are aligned to their size)
KVZ_HD bool is_cu_edge_v(const EncFrame &f, int x, int y) { return (x & ((1 << f.cu_log2[b8idx(f, x, y)]) - 1)) == 0; }
KVZ_HD bool is_cu_edge_h(const EncFrame &f, int x, int y) { return (y & ((1 << f.cu_log2[b8idx(f, x, y)]) - 1)) == 0; }

// ---------------------------------------------------------------------------------------------
// scalar quantiser / dequantiser (flat scaling; see oracle/hevc_transform.c for the statement)
// ---------------------------------------------------------------------------------------------
// m: the scaling factor of the coefficient's position (8.6.4.2; `scaling-list default`), 16 = flat.  The forward scale of a position is the flat one times
// 16 / m, as Kvazaar's (and HM's) quantisation matrices are built: (quantScales[qp % 6] << 4) / m.  Statement: oracle/hevc_transform.c orc_quant.
KVZ_HD int quant_scale_m(int qp, int m) { return (kQuantScale[qp % 6] << 4) / m; }
KVZ_HD int quant_level(int coef, int qp, int log2n, int intra, int m = 16)
{
  int shift = 14 + qp / 6 + (15 - 8 - log2n);
  int64_t off = (int64_t)(intra ? 171 : 85) << (shift - 9);
  int a = coef < 0 ? -coef : coef;
  int64_t q = ((int64_t)a * quant_scale_m(qp, m) + off) >> shift;
  if (q > 32767) q = 32767;
  return (int)(coef < 0 ? -q : q);
}
// The quantiser with what the level-adjustment pass needs beside the level: aux = 256 + du in bits 0..9 -- du = the part of the coefficient
// the level does not account for, in 1/256 quantiser steps -- and bit 15 = the coefficient is negative (statement: oracle/hevc_transform.h)
KVZ_HD int quant_level_aux(int coef, int qp, int log2n, int intra, uint16_t *aux, int m = 16)
{
  int shift = 14 + qp / 6 + (15 - 8 - log2n);
  int64_t off = (int64_t)(intra ? 171 : 85) << (shift - 9);
  int a = coef < 0 ? -coef : coef;
  int64_t prod = (int64_t)a * quant_scale_m(qp, m);
  int64_t q = (prod + off) >> shift;
  if (q > 32767) q = 32767;
  int64_t du = (prod >> (shift - 8)) - (q << 8);
  if (du < -256) du = -256;
  if (du > 511) du = 511;
  *aux = (uint16_t)((du + 256) | (coef < 0 ? 0x8000 : 0));
  return (int)(coef < 0 ? -q : q);
}

// "uvgx RDOQ v1" and sign data hiding on ONE 4x4 coefficient group (statement of record: orc_adjust_levels, oracle/hevc_transform.h): lv[k], aux[k]
// = level and quantiser remainder at scan position k of the group, dc_group = it is the block's first group.  Returns the non-zero levels left.
KVZ_HD int adjust_group(int16_t (&lv)[16], const uint16_t (&aux)[16], bool dc_group, int rdoq, int signhide)
{
  int nz = 0; bool ones = true;
  for (int k = 0; k < 16; k++) if (lv[k]) { nz++; if (lv[k] != 1 && lv[k] != -1) ones = false; }
  if (rdoq && !dc_group && nz >= 1 && nz <= 2 && ones) {
    int benefit = 0;
    for (int k = 0; k < 16; k++) if (lv[k]) benefit += 2 * (int)(aux[k] & 0x3ff) - 256;      // 2 u - 256, u = du + 256
    if (benefit < 92 * nz + 92) { for (int k = 0; k < 16; k++) lv[k] = 0; nz = 0; }
  }
  if (signhide && nz >= 2) {
    int first = -1, last = -1, sum = 0;
    for (int k = 0; k < 16; k++) if (lv[k]) { if (first < 0) first = k; last = k; sum += lv[k] < 0 ? -lv[k] : lv[k]; }
    if (last - first >= 4 && (sum & 1) != (lv[first] < 0 ? 1 : 0)) {
      int best = -1, best_cost = 1 << 30, best_change = 0;
      for (int k = 15; k >= 0; k--) {
        const int a = lv[k] < 0 ? -lv[k] : lv[k], du = (int)(aux[k] & 0x3ff) - 256;
        int up, down; bool ok_up, ok_down;
        if (a) { ok_up = a < 32767; up = 256 - 2 * du + 23; ok_down = !(a == 1 && (k == first || k == last)); down = 256 + 2 * du - (a == 1 ? 69 : 23); }
        else { ok_up = k > first; up = 256 - 2 * du + 80; ok_down = false; down = 0; }
        if (ok_up && up < best_cost) { best_cost = up; best = k; best_change = 1; }
        if (ok_down && down < best_cost) { best_cost = down; best = k; best_change = -1; }
      }
      if (best >= 0) {
        const int a = lv[best] < 0 ? -lv[best] : lv[best];
        if (a == 0) { lv[best] = (int16_t)((aux[best] >> 15) ? -1 : 1); nz++; }
        else { const int na = a + best_change; lv[best] = (int16_t)(lv[best] < 0 ? -na : na); if (na == 0) nz--; }
      }
    }
  }
  return nz;
}

// ... with the scaling factor m of the coefficient's position (scaling lists); m = 16 is the flat case below
KVZ_HD int dequant_coef_m(int level, int qp, int log2n, int m)
{
  int bd = 8 + log2n - 5;
  int scale = kLevelScale[qp % 6] << (qp / 6);
  int64_t v = ((int64_t)level * m * scale + ((int64_t)1 << (bd - 1))) >> bd;
  return (int)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}
KVZ_HD int dequant_coef(int level, int qp, int log2n, int m)      // (the encoder's call sites: m = 16 without scaling lists)
{
  return dequant_coef_m(level, qp, log2n, m);
}
KVZ_HD int dequant_coef(int level, int qp, int log2n)
{
  int bd = 8 + log2n - 5;
  int scale = kLevelScale[qp % 6] << (qp / 6);
  int64_t v = ((int64_t)level * 16 * scale + ((int64_t)1 << (bd - 1))) >> bd;
  return (int)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}

}  // namespace kvzx
