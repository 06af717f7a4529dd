// kvazzup_amd/csrc/dec_frame.h -- what the host half of the decoder (CABAC parse, decoder.hip) hands to the device half
// (dec_kernels.hip) for one picture.  The decoder sits behind libOpenHevcDecode, which uvgComm feeds with any peer's
// stream (/root/reference/src/media/processing/openhevcfilter.cpp:134-172), so the layout is general for Main-profile
// I / P / B pictures: any coding quadtree with CTB 64 and minimum CB 8, every partitioning, transform trees down to 4x4,
// several reference pictures, coded sizes that are multiples of 8.
//
// Everything lives in ONE pinned host block per picture that goes to the GPU in one copy:
//   [ B4Rec x (ph/4 * pw/4) | region table | CTU table | SaoParams x CTUs | DecTu x ntu | level words ]
#pragma once
#include <stdint.h>
#include "hevc_core.h"

namespace kvzx {

// per 4x4 luma block (raster, pitch pw / 4): motion for prediction, and what deblocking needs to derive the boundary strength
// (8.7.2.4) -- transform / prediction block edges on the 8x8 grid, "transform block has coefficients", QpY
struct B4Rec {
  int16_t mvx, mvy;        // quarter luma samples
  int8_t ref_idx;          // index into RefPicList0; -1 = intra (host: also what merge candidates compare, 8.5.3.2.3).  B slices: >= 0 = inter, the vector and
                           // slot are those of list 0 when it is used, else list 1's (a uni-predicted block is the same to the kernels either way)
  uint8_t flags;           // B4_*
  int8_t qp_y;             // QpY of the coding unit (8.6.1)
  uint8_t slot;            // picture buffer of the reference picture: two indices naming one picture share it (8.7.2.4 compares pictures)
};
// the SECOND motion of a bi-predicted block (B4_BI; B slices): the list-1 vector and its picture buffer.  An array of its own beside b4[], present
// only in pictures that hold such a block -- P pictures, every picture of a default Kvazaar peer, never pay for it
struct B4L1 { int16_t mvx, mvy; uint8_t slot; uint8_t pad[3]; };      // pad[0], pad[1]: B4_WT -- entries of the weight table (DecWt: list * 16 + index) of the first / the second motion
// explicit weighted prediction (pred_weight_table(), 7.4.7.3): weight and offset per list, reference index and colour component
struct DecWt { int16_t w[3], o[3]; };
enum { B4_WT = 128,        // a slice with pred_weight_table(): b4x[] holds the block's weight table entries (and the second vector when B4_BI is set too)
       B4_BI = 64,          // bi-predicted: b4x[] holds the second vector (B4Rec: list 0's)
       B4_BYPASS = 32,      // cu_transquant_bypass_flag: the loop filters leave this block's samples as they are (8.7.2.5.7, 8.7.3)
       B4_NZ = 1,          // the luma transform block covering this 4x4 has non-zero coefficients
       B4_EDGE_V = 2,      // the left edge of this 4x4 is a transform-block or prediction-block edge
       B4_TU_V = 4,        //   ... a transform-block edge
       B4_EDGE_H = 8, B4_TU_H = 16 };      // the same for the top edge

// one transform block, in decoding order.  Intra blocks are listed even without coefficients (count == 0): they are the unit
// of intra prediction (8.4.4.1).  x, y in samples of the block's own plane.
struct DecTu {
  uint16_t x, y;
  uint8_t plane, log2;
  uint8_t flags;           // TU_*
  uint8_t mode;            // intra prediction mode of this block (chroma: already derived, 8.4.3)
  int8_t qp;               // quantisation parameter of this block's plane (qP of 8.6.2: Qp'Y, Qp'Cb or Qp'Cr)
  uint8_t pad;
  uint16_t count;          // non-zero levels: `count` words (raster position inside the block << 16 | level & 0xffff) from word `offset`
  uint32_t offset;
};
enum { TU_INTRA = 1, TU_TSKIP = 2, TU_DST = 4, TU_BYPASS = 8 };      // TU_BYPASS: cu_transquant_bypass_flag -- the residual is the level array itself (8.6.2)

struct TuRange { uint32_t first, count; };

#define KVZ_DEC_MAX_REFS 16

// kernel argument of every decoder kernel
struct DecFrame {
  int w, h;                 // coded luma size (multiples of 8)
  int pw, ph;               // luma pitch and allocated rows (multiples of 64); chroma planes pw / 2 x ph / 2
  int wc, hc;               // 64x64 tiles of the picture per row / column, partial ones included: the grids of the region, deblocking and SAO kernels (= the CTUs of a stream with 64x64 CTBs)
  int ctb_log2, cwc, chc;   // CtbLog2SizeY (6, 5 or 4) and the picture in CODING TREE BLOCKS: the intra chain's work units, ctu[], ctu_tile[], sao[], the edge words (round 6; cwc == wc, chc == hc at 64)
  const uint32_t *tu_index; // CTBs smaller than 64: region[] names a run of this list of indices into tus[] (a region's blocks are no run of tus[] there); NULL: a run of tus[] itself
  int row0, nrows;          // band of CTU rows the launch works on (nrows == 0: the whole picture): tile-row split over several decoders
  const B4Rec *b4;          // [ph / 4][pw / 4]
  const B4L1 *b4x;          // [ph / 4][pw / 4] second vectors of the B4_BI blocks (weight entries of the B4_WT blocks), NULL: the picture has none
  const DecWt *wt; uint8_t wt_log2[2];      // explicit weighted prediction: 32 entries (list * 16 + index), luma / chroma denominators; NULL: default weights
  const TuRange *region;      // per 32x32 luma region (raster, pitch 2 * wc): {first transform block, count} in tus[]
  const TuRange *ctu;        // per CTB (raster, pitch cwc): {first transform block, count | intra-planes mask << 24}
  const DecTu *tus; const uint32_t *lev;
  int ntu;                  // transform blocks in tus[]
  int16_t *resid[3];        // residual of the intra transform blocks up to 16x16 (k_dec_intra_resid -> k_dec_intra), plane-shaped like rec[]
  const uint8_t *ctu_tile;  // per CTB (raster, pitch cwc): (tile row mod 16) << 4 | (tile column mod 16) -- distinct for adjacent tiles, which is all the kernels compare
  uint8_t *rec[3];          // the picture being reconstructed (and deblocked in place)
  uint8_t *out[3];          // SAO output (== rec planes of the picture buffer when SAO runs from a work picture)
  const uint8_t *ref[KVZ_DEC_MAX_REFS][3];   // picture buffers by slot
  const SaoParams *sao;     // per CTB (raster, pitch cwc); NULL = off
  const uint8_t *ctu_nb;    // per CTB (raster, pitch cwc), NULL = no boundary inside the picture is closed to the in-loop filters: bit k = the samples of the neighbouring CTB k
                            // (NW N NE W E SW S SE = 0 .. 7) may be used -- a slice with slice_loop_filter_across_slices_enabled_flag = 0, tiles with
                            // loop_filter_across_tiles_enabled_flag = 0 (Kvazaar's).  SAO looks here; for deblocking the host has taken the edge marks off the records
  // k_dec_intra's hand-off between CTUs (kernel_common.h IntraNeighbours): per plane and CTU the right column / bottom row of its intra blocks as
  // self-validating words -- one sample | chain_gen << 8 per row, four samples | chain_gen << 32 per four columns -- that the neighbouring CTU's wave
  // polls until they carry this launch's generation
  uint32_t *edge_col[3]; unsigned long long *edge_row[3]; uint32_t chain_gen;
  uint32_t *progress;       // k_dec_intra's ticket counter, at [3 * CTUs] (the per-CTU progress counters in front of it are no longer used)
  const uint32_t *intra_order;  // CTU handled by the k-th workgroup triple of k_dec_intra: anti-diagonal order (enc_kernels.hip k_intra_recon)
  uint32_t *err;
  int8_t cb_qp_offset, cr_qp_offset;       // pps_cb/cr_qp_offset (deblocking uses these, 8.7.2.5.5)
  int8_t beta_offset, tc_offset;           // slice_beta_offset_div2 * 2, slice_tc_offset_div2 * 2
  uint8_t intra_direct;     // k_dec_intra: a workgroup per (CTU, plane) in dispatch order instead of tickets (pictures that are mostly inter: nearly every (CTU, plane) has nothing to do, and a ticket is a memory round trip)
  uint8_t strong_intra, tiles;             // strong_intra_smoothing_enabled_flag; more than one tile
  uint8_t general;                         // the picture may hold what the chain's plain form has no code for: several slices or tiles (reference samples in two runs), PCM units
  uint8_t cip;                             // constrained_intra_pred_flag: neighbouring samples of blocks that are not intra-coded are "not available" as reference samples (8.4.4.2.2)
  uint8_t tq_bypass;        // the picture may hold coding units with cu_transquant_bypass_flag (B4_BYPASS / TU_BYPASS): the loop filters look at the flags
  const uint8_t *scaling;   // KVZ_SCALING_BYTES scaling factors (scaling_list_enabled_flag), NULL: flat 16
};

// Several pictures in ONE launch (batch.h: pictures of different decoder instances that are ready at the same time): the kernel argument is
// a table of frame descriptors in DEVICE memory -- each picture's DecFrame travels at the end of the fixed part of its input block -- and
// the workgroups' share-out: frame i owns workgroups [first[i], first[i + 1]) of the one-dimensional grid (every share a multiple of 8, so
// that a workgroup's index inside its share still names its XCD; the padding workgroups leave at once).
#define KVZ_DEC_BATCH_MAX 8
struct DecBatch {
  const DecFrame *f[KVZ_DEC_BATCH_MAX];
  uint32_t first[KVZ_DEC_BATCH_MAX + 1];      // unused entries = first[n]
  uint32_t count[KVZ_DEC_BATCH_MAX];          // workgroups of frame i that have work (<= its share)
};

// z-order index (0..63) of the 8x8 luma unit at (xi, yi) of a CTU, xi, yi in 0..7
KVZ_HD int zunit8(int xi, int yi)
{
  int z = 0;
  for (int b = 0; b < 3; b++) z |= ((xi >> b) & 1) << (2 * b) | ((yi >> b) & 1) << (2 * b + 1);
  return z;
}

}  // namespace kvzx
