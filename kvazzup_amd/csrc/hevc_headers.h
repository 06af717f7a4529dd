// kvazzup_amd/csrc/hevc_headers.h -- host-side bit writer, parameter sets (H.265 7.3.2), slice
// segment header (7.3.6.1) and NAL framing (7.3.1.1, Annex B) for the encoder's fixed tool set.
// Replaces the bitstream assembly Kvazaar does at the end of encoder_encode
// (/root/reference/src/media/processing/kvazaarfilter.cpp:435-438,469-474 consume its chunks).
#pragma once
#include <stdint.h>
#include <cstring>
#include <vector>

namespace kvzx {

class BitWriter {
 public:
  void put(uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) bit((v >> i) & 1u); }
  void bit(uint32_t b) { cur_ = (cur_ << 1) | (b & 1u); if (++n_ == 8) { buf_.push_back((uint8_t)cur_); cur_ = 0; n_ = 0; } }
  void ue(uint32_t v) { uint32_t x = v + 1; int len = 0; while ((x >> len) > 1) len++; put(0, len); put(x, len + 1); }
  void se(int32_t v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
  void trailing() { bit(1); while (n_) bit(0); }
  bool aligned() const { return n_ == 0; }
  void bytes(const uint8_t *p, size_t n) { buf_.insert(buf_.end(), p, p + n); }
  std::vector<uint8_t> &data() { return buf_; }
 private:
  std::vector<uint8_t> buf_; uint32_t cur_ = 0; int n_ = 0;
};

struct StreamParams {
  int cw, ch;               // coded size
  int width, height;        // display size (conformance window)
  int qp, wpp, deblock;
  int fps_num, fps_den;
  int qp_in_cu = 0;         // cu_qp_delta_enabled_flag with diff_cu_qp_delta_depth 0 (quantisation group = CTU)
  int tile_rows = 1;        // tile_rows * tile_cols > 1: tiles_enabled_flag, uniform spacing, loop filter across tiles on
  int tile_cols = 1;
  int sao = 0;              // sample_adaptive_offset_enabled_flag; every slice: slice_sao_luma_flag = slice_sao_chroma_flag = 1
  int signhide = 0;         // sign_data_hiding_enabled_flag
  int scaling_list = 0;     // scaling_list_enabled_flag = 1, no sps_scaling_list_data: the default lists
  int tq_bypass = 0;        // transquant_bypass_enabled_flag (`lossless`: every coding unit sets cu_transquant_bypass_flag)
  int slices = 0;           // kvazaar slices: 1 = "wpp", a dependent slice segment per CTU row (dependent_slice_segments_enabled_flag); 2 = "tiles", a slice per tile
};

inline int level_idc_for(int w, int h)
{
  long px = (long)w * h;
  return px <= 2228224 ? 123 : (px <= 8912896 ? 153 : 183);
}

inline void write_ptl(BitWriter &w, int level)
{
  w.put(0, 2); w.put(0, 1); w.put(1, 5);                         // profile space, tier, Main
  for (int j = 0; j < 32; j++) w.bit(j == 1 || j == 2);
  w.bit(1); w.bit(0); w.bit(0); w.bit(1);                        // progressive, interlaced, non-packed, frame-only
  w.put(0, 32); w.put(0, 11); w.bit(0);
  w.put((uint32_t)level, 8);
}

inline void write_vps(BitWriter &w, const StreamParams &s)
{
  w.put(0, 4); w.put(3, 2); w.put(0, 6); w.put(0, 3); w.bit(1); w.put(0xffff, 16);
  write_ptl(w, level_idc_for(s.cw, s.ch));
  w.bit(1); w.ue(1); w.ue(0); w.ue(0);                           // ordering info: dpb 2, no reorder
  w.put(0, 6); w.ue(0);
  w.bit(1); w.put((uint32_t)s.fps_den, 32); w.put((uint32_t)s.fps_num, 32); w.bit(0); w.ue(0);
  w.bit(0);
  w.trailing();
}

inline void write_sps(BitWriter &w, const StreamParams &s)
{
  w.put(0, 4); w.put(0, 3); w.bit(1);
  write_ptl(w, level_idc_for(s.cw, s.ch));
  w.ue(0); w.ue(1); w.ue((uint32_t)s.cw); w.ue((uint32_t)s.ch);
  bool crop = s.cw != s.width || s.ch != s.height;
  w.bit(crop);
  if (crop) { w.ue(0); w.ue((uint32_t)(s.cw - s.width) / 2); w.ue(0); w.ue((uint32_t)(s.ch - s.height) / 2); }
  w.ue(0); w.ue(0); w.ue(4);                                     // 8-bit, log2_max_poc_lsb 8
  w.bit(1); w.ue(1); w.ue(0); w.ue(0);
  w.ue(0); w.ue(3); w.ue(0); w.ue(3);                            // CB 8..64, TB 4..32
  w.ue(0); w.ue(0);                                              // transform hierarchy depths
  w.bit(s.scaling_list != 0); if (s.scaling_list) w.bit(0);       // scaling_list_enabled_flag (sps_scaling_list_data_present_flag = 0: Tables 7-5 / 7-6)
  w.bit(0); w.bit(s.sao != 0); w.bit(0);                          // amp, sao, pcm
  w.ue(1); w.ue(1); w.ue(0); w.ue(0); w.bit(1);                  // one short-term RPS: previous picture
  w.bit(0); w.bit(0); w.bit(1);                                  // long-term, tmvp, strong intra smoothing
  w.bit(1);                                                      // VUI: timing only
  w.put(0, 8);
  w.bit(1); w.put((uint32_t)s.fps_den, 32); w.put((uint32_t)s.fps_num, 32); w.bit(0); w.bit(0);
  w.bit(0);
  w.bit(0);
  w.trailing();
}

inline void write_pps(BitWriter &w, const StreamParams &s)
{
  w.ue(0); w.ue(0);
  w.bit(s.slices == 1); w.bit(0); w.put(0, 3); w.bit(s.signhide != 0); w.bit(0);   // dependent_slice_segments_enabled_flag, output_flag_present, extra header bits, sign_data_hiding_enabled_flag, cabac_init_present
  w.ue(0); w.ue(0);
  w.se(s.qp - 26);
  w.bit(0); w.bit(0); w.bit(s.qp_in_cu != 0);                    // constrained intra, transform skip, cu_qp_delta
  if (s.qp_in_cu) w.ue(0);                                       // diff_cu_qp_delta_depth
  w.se(0); w.se(0); w.bit(0);
  w.bit(0); w.bit(0); w.bit(s.tq_bypass != 0);                  // weighted_pred, weighted_bipred, transquant_bypass_enabled
  w.bit(s.tile_rows > 1 || s.tile_cols > 1); w.bit(s.wpp);       // tiles, entropy_coding_sync
  if (s.tile_rows > 1 || s.tile_cols > 1) { w.ue((uint32_t)s.tile_cols - 1); w.ue((uint32_t)s.tile_rows - 1); w.bit(1); w.bit(1); }   // columns - 1, rows - 1, uniform spacing, loop filter across tiles
  w.bit(1);                                                      // loop filter across slices
  w.bit(!s.deblock);
  if (!s.deblock) { w.bit(0); w.bit(1); }
  w.bit(0); w.bit(0); w.ue(0); w.bit(0); w.bit(0);
  w.trailing();
}

// Emulation prevention (7.4.2): arithmetic-coded data holds a zero byte every ~256 bytes, so the bytes between zero bytes are found with memchr and moved
// in one piece, and only the zero runs (and the byte behind them) go through the byte-by-byte rule (round 5: these loops and the decoder's inverse were
// ~3 % of the host's time at a byte per iteration).
inline size_t escaped_size(const uint8_t *p, size_t n)
{
  size_t out = n, i = 0; int zeros = 0;
  while (i < n) {
    if (zeros == 0) {
      const uint8_t *z = (const uint8_t *)memchr(p + i, 0, n - i);
      if (!z) break;
      i = (size_t)(z - p);
    }
    if (zeros >= 2 && p[i] <= 3) { out++; zeros = 0; }
    zeros = p[i] == 0 ? zeros + 1 : 0; i++;
  }
  return out;
}

// slice segment header (7.3.6.1).  address < 0: the picture's first segment; else slice_segment_address of a further one, `dependent`
// = a dependent slice segment (nothing but the address and the entry points)
inline void write_slice_header(BitWriter &w, const StreamParams &s, bool idr, int poc, const std::vector<uint32_t> &entry_sizes, int slice_qp_delta = 0,
                               int address = -1, bool dependent = false)
{
  w.bit(address < 0);
  if (idr) w.bit(0);
  w.ue(0);
  if (address >= 0) {
    if (s.slices == 1) w.bit(dependent);
    const int nctb = (s.cw / 64) * (s.ch / 64);
    int bits = 0; while ((1 << bits) < nctb) bits++;
    w.put((uint32_t)address, bits);
  }
  if (!dependent) {
    w.ue(idr ? 2 : 1);
    if (!idr) { w.put((uint32_t)poc & 255, 8); w.bit(1); }
    if (s.sao) { w.bit(1); w.bit(1); }                             // slice_sao_luma_flag, slice_sao_chroma_flag
    if (!idr) { w.bit(0); w.ue(0); }                               // num_ref_idx override, five_minus_max_num_merge_cand
    w.se(slice_qp_delta);                                          // against the PPS init_qp (= the configured QP)
    // (deblocking override not enabled; slice_loop_filter_across_slices_enabled_flag present when deblocking or SAO is on)
    if (s.deblock || s.sao) w.bit(1);
  }
  if (s.wpp || s.tile_rows > 1 || s.tile_cols > 1) {
    w.ue((uint32_t)entry_sizes.size());
    if (!entry_sizes.empty()) {
      uint32_t mx = 0; for (uint32_t e : entry_sizes) if (e - 1 > mx) mx = e - 1;
      int len = 1; while (len < 32 && (mx >> len)) len++;
      w.ue((uint32_t)len - 1);
      for (uint32_t e : entry_sizes) w.put(e - 1, len);
    }
  }
  w.trailing();
}

// Append a NAL unit with a 4-byte start code and emulation prevention.
inline void append_nal(std::vector<uint8_t> &out, int nal_type, const uint8_t *rbsp, size_t n)
{
  out.push_back(0); out.push_back(0); out.push_back(0); out.push_back(1);
  out.push_back((uint8_t)(nal_type << 1)); out.push_back(1);
  out.reserve(out.size() + n + n / 128 + 16);
  int zeros = 0; size_t i = 0;
  while (i < n) {
    if (zeros == 0) {                                       // up to the next zero byte in one piece
      const uint8_t *z = (const uint8_t *)memchr(rbsp + i, 0, n - i);
      const size_t k = z ? (size_t)(z - (rbsp + i)) : n - i;
      out.insert(out.end(), rbsp + i, rbsp + i + k); i += k;
      if (i >= n) break;
    }
    if (zeros >= 2 && rbsp[i] <= 3) { out.push_back(3); zeros = 0; }
    out.push_back(rbsp[i]);
    zeros = rbsp[i] == 0 ? zeros + 1 : 0; i++;
  }
}

// One access unit: [VPS SPS PPS] + slice NAL whose data are the `nsub` substreams (CTU rows with
// WPP, otherwise one) rows[r].  false (and an empty access unit): the substream count does not fit the tiling.
inline bool assemble_access_unit(std::vector<uint8_t> &au, const StreamParams &sp, bool idr, int poc, bool write_ps,
                                 const std::vector<std::vector<uint8_t>> &rows, int nsub, int slice_qp_delta = 0)
{
  au.clear();
  if (write_ps) {
    BitWriter a, b, c;
    write_vps(a, sp); append_nal(au, 32, a.data().data(), a.data().size());
    write_sps(b, sp); append_nal(au, 33, b.data().data(), b.data().size());
    write_pps(c, sp); append_nal(au, 34, c.data().data(), c.data().size());
  }
  // slice segments, one NAL unit each: the whole picture; or (slices 1, WPP) a dependent slice segment per CTU row; or (slices 2) an
  // independent slice per tile -- the tile's CTU rows with WPP, else its one substream
  const int hc = sp.ch / 64, wc = sp.cw / 64;
  // the substreams come in decoding order (tile scan): for each, whether it starts a tile and its first CTB
  std::vector<int> tile_first, addr_of;
  for (int tr = 0; tr < sp.tile_rows; tr++)
    for (int tc = 0; tc < sp.tile_cols; tc++) {
      const int cy0 = tile_row_first(hc, sp.tile_rows, tr), cy1 = tile_row_first(hc, sp.tile_rows, tr + 1), cx0 = tile_col_first(wc, sp.tile_cols, tc);
      for (int cy = cy0; cy < (sp.wpp ? cy1 : cy0 + 1); cy++) { tile_first.push_back(cy == cy0); addr_of.push_back(cy * wc + cx0); }
    }
  if ((int)tile_first.size() != nsub) { au.clear(); return false; }   // the caller's substream count does not fit the tiling: no access unit rather than one without a slice
  for (int s0 = 0; s0 < nsub;) {
    int n = nsub - s0, addr = s0 ? addr_of[(size_t)s0] : -1;
    if (sp.slices == 1) n = 1;
    else if (sp.slices == 2) { n = 1; while (s0 + n < nsub && !tile_first[(size_t)(s0 + n)]) n++; }
    std::vector<uint32_t> entry;
    for (int r = 0; r + 1 < n; r++) entry.push_back((uint32_t)escaped_size(rows[(size_t)(s0 + r)].data(), rows[(size_t)(s0 + r)].size()));
    BitWriter sh;
    write_slice_header(sh, sp, idr, poc, entry, slice_qp_delta, addr, sp.slices == 1 && s0 > 0);
    for (int r = 0; r < n; r++) sh.bytes(rows[(size_t)(s0 + r)].data(), rows[(size_t)(s0 + r)].size());
    append_nal(au, idr ? 19 : 1, sh.data().data(), sh.data().size());
    s0 += n;
  }
  return true;
}

}  // namespace kvzx
