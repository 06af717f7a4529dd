// kvazzup_amd/csrc/openhevc_api.hip -- the libOpenHevc* C ABI (include/openHevcWrapper.h) and the
// decoder extensions (include/kvazzup_amd.h) on top of kvzx::Decoder.
// Drop-in for the calls uvgComm makes at /root/reference/src/media/processing/openhevcfilter.cpp:
// 36-56 (init), 145-146 (decode), 195-199 (output), 81-82 (flush/close).
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../../include/kvazzup_amd.h"
#include "decoder.h"

using kvzx::Decoder;
using kvzx::DecodedPicture;

namespace {
struct Handle {
  Decoder *dec = nullptr;
  int threads = 1, thread_type = 0;
  bool started = false;
  DecodedPicture pic; bool have_pic = false;
  const uint8_t *planes[3] = {nullptr, nullptr, nullptr};
};
Handle *H(OpenHevc_Handle h) { return (Handle *)h; }
void fill_info(const DecodedPicture &p, OpenHevc_FrameInfo *info, bool host)
{
  memset(info, 0, sizeof(*info));
  info->nYPitch = host ? p.host_pitch[0] : p.dev_pitch[0];
  info->nUPitch = host ? p.host_pitch[1] : p.dev_pitch[1];
  info->nVPitch = host ? p.host_pitch[2] : p.dev_pitch[2];
  info->nBitDepth = 8; info->nWidth = p.width; info->nHeight = p.height; info->chromat_format = YUV420;
  info->sample_aspect_ratio.num = 1; info->sample_aspect_ratio.den = 1;
  info->frameRate.num = (int)p.fps_num; info->frameRate.den = (int)p.fps_den;
  info->display_picture_number = p.poc; info->flag = 0; info->nTimeStamp = p.pts;
}
}  // namespace

extern "C" {

OpenHevc_Handle libOpenHevcInit(int nb_pthreads, int thread_type)
{
  Handle *h = new Handle();
  h->threads = nb_pthreads; h->thread_type = thread_type;       // accepted; the GPU does the sample work
  const char *env = getenv("KVAZZUP_AMD_DEVICE");
  h->dec = new Decoder(env ? atoi(env) : 0);
  if ((thread_type & 1) && nb_pthreads > 1) h->dec->set_frame_threads(nb_pthreads);   // OH_THREAD_FRAME / OH_THREAD_FRAMESLICE
  return (OpenHevc_Handle)h;
}
int libOpenHevcStartDecoder(OpenHevc_Handle hh)
{
  Handle *h = H(hh);
  if (!h) return -1;
  std::string err;
  if (!h->dec->start(&err)) { fprintf(stderr, "kvazzup_amd: libOpenHevcStartDecoder failed: %s\n", err.c_str()); return -1; }
  h->started = true;
  return 0;
}
int libOpenHevcDecode(OpenHevc_Handle hh, const unsigned char *buff, int nal_len, int64_t pts)
{
  Handle *h = H(hh);
  if (!h || !h->started || !buff || nal_len <= 0) return -1;
  h->have_pic = false;
  int rc = h->dec->decode_nal(buff, (size_t)nal_len, pts);
  if (rc > 0) h->have_pic = h->dec->get_picture(&h->pic);
  return rc;
}
int libOpenHevcGetOutput(OpenHevc_Handle hh, int got_picture, OpenHevc_Frame *frame)
{
  Handle *h = H(hh);
  if (!h || !frame || got_picture <= 0 || !h->have_pic) return 0;
  frame->pvY = (void **)h->pic.host[0]; frame->pvU = (void **)h->pic.host[1]; frame->pvV = (void **)h->pic.host[2];
  fill_info(h->pic, &frame->frameInfo, true);
  return 1;
}
int libOpenHevcGetOutputCpy(OpenHevc_Handle hh, int got_picture, OpenHevc_Frame_cpy *frame)
{
  Handle *h = H(hh);
  if (!h || !frame || got_picture <= 0 || !h->have_pic || !h->pic.host[0]) return 0;
  const DecodedPicture &p = h->pic;
  void *dst[3] = {frame->pvY, frame->pvU, frame->pvV};
  for (int c = 0; c < 3; c++) {
    int w = c ? p.width / 2 : p.width, hh2 = c ? p.height / 2 : p.height;
    if (!dst[c]) return 0;
    for (int y = 0; y < hh2; y++) memcpy((uint8_t *)dst[c] + (size_t)y * w, p.host[c] + (size_t)y * p.host_pitch[c], (size_t)w);
  }
  fill_info(p, &frame->frameInfo, true);
  return 1;
}
void libOpenHevcGetPictureInfo(OpenHevc_Handle hh, OpenHevc_FrameInfo *info)
{
  Handle *h = H(hh);
  if (!h || !info) return;
  if (h->have_pic) fill_info(h->pic, info, true); else memset(info, 0, sizeof(*info));
}
void libOpenHevcGetPictureSize2(OpenHevc_Handle hh, OpenHevc_FrameInfo *info) { libOpenHevcGetPictureInfo(hh, info); }
void libOpenHevcSetCheckMD5(OpenHevc_Handle hh, int val) { Handle *h = H(hh); if (h && h->dec) h->dec->set_check_hash(val != 0); }
void libOpenHevcSetDebugMode(OpenHevc_Handle, int) {}
void libOpenHevcSetTemporalLayer_id(OpenHevc_Handle hh, int val) { Handle *h = H(hh); if (h && h->dec) h->dec->set_max_temporal_id(val); }
void libOpenHevcSetNoCropping(OpenHevc_Handle hh, int val) { Handle *h = H(hh); if (h && h->dec) h->dec->set_no_cropping(val != 0); }
void libOpenHevcSetActiveDecoders(OpenHevc_Handle, int) {}
void libOpenHevcSetViewLayers(OpenHevc_Handle, int) {}
void libOpenHevcFlush(OpenHevc_Handle hh) { Handle *h = H(hh); if (h && h->dec) h->dec->flush(); }
void libOpenHevcClose(OpenHevc_Handle hh)
{
  Handle *h = H(hh);
  if (!h) return;
  delete h->dec;
  delete h;
}
const char *libOpenHevcVersion(OpenHevc_Handle) { return "kvazzup_amd-hevc-dec 0.1 (gfx950)"; }

int kvzx_decoder_set_device(OpenHevc_Handle hh, int device)
{
  Handle *h = H(hh);
  if (!h || h->started || device < 0) return 0;
  int ft = h->dec->frame_threads();
  delete h->dec;
  h->dec = new Decoder(device);
  h->dec->set_frame_threads(ft);
  return 1;
}
void kvzx_decoder_hash_stats(OpenHevc_Handle hh, int *checked, int *mismatch) { Handle *h = H(hh); if (h) h->dec->hash_stats(checked, mismatch); }
// test / measurement hook (decoder.h set_parse_only): before libOpenHevcStartDecoder; the decoder then parses and never outputs a picture
int kvzx_decoder_set_parse_only(OpenHevc_Handle hh, int parse_threads)
{
  Handle *h = H(hh);
  if (!h || h->started) return 0;
  if (!h->dec->set_parse_only()) return 0;                // (the decoder declined: it has been started)
  h->dec->set_parse_threads(parse_threads < 1 ? 1 : parse_threads);
  return 1;
}
int kvzx_decoder_parse_probe_stats(OpenHevc_Handle hh, uint64_t *out5, double *parse_ms)
{
  Handle *h = H(hh);
  if (!h) return 0;
  const Decoder::ProbeStats st = h->dec->parse_probe_stats();
  if (out5) { out5[0] = st.pictures; out5[1] = st.tus; out5[2] = st.levels; out5[3] = st.digest; out5[4] = st.bins; }
  if (parse_ms) *parse_ms = st.parse_ms;
  return 1;
}
int kvzx_decoder_last_error(OpenHevc_Handle hh) { Handle *h = H(hh); return h ? h->dec->last_error() : -1; }
int kvzx_decoder_output_device(OpenHevc_Handle hh, const void **planes, int *pitches)
{
  Handle *h = H(hh);
  if (!h || !h->have_pic) return 0;
  for (int c = 0; c < 3; c++) { if (planes) planes[c] = h->pic.dev[c]; if (pitches) pitches[c] = h->pic.dev_pitch[c]; }
  return 1;
}
int kvzx_decoder_output_rgb32_device(OpenHevc_Handle hh, void *d_rgb32, int variant)
{
  Handle *h = H(hh);
  if (!h || !h->have_pic || !d_rgb32) return 0;
  const DecodedPicture &p = h->pic;
  if (!kvzx_yuv420_to_rgb32_device(p.dev[0], p.dev[1], p.dev[2], p.dev_pitch[0], p.dev_pitch[1], d_rgb32, p.width, p.height, variant, nullptr)) return 0;
  return hipStreamSynchronize(nullptr) == hipSuccess ? 1 : 0;
}
// ---- tile-row split decoder (decoder.h "band mode")
int kvzx_decoder_set_band(OpenHevc_Handle hh, int row0, int nrows) { Handle *h = H(hh); if (!h || row0 < 0 || nrows < 0) return 0; h->dec->set_band(row0, nrows); return 1; }
size_t kvzx_decoder_band_halo_bytes(OpenHevc_Handle hh) { Handle *h = H(hh); return h ? h->dec->band_halo_bytes() : 0; }
int kvzx_decoder_band_export(OpenHevc_Handle hh, int stage, void *d_buf) { Handle *h = H(hh); return h && h->dec->band_export(stage, (uint8_t *)d_buf) ? 1 : 0; }
int kvzx_decoder_band_import(OpenHevc_Handle hh, int stage, const void *d_buf) { Handle *h = H(hh); return h && h->dec->band_import(stage, (const uint8_t *)d_buf) ? 1 : 0; }
int kvzx_decoder_band_ready(OpenHevc_Handle hh) { Handle *h = H(hh); return h && h->dec->band_ready() ? 1 : 0; }
int kvzx_decoder_band_deblock(OpenHevc_Handle hh) { Handle *h = H(hh); return h && h->dec->band_deblock() ? 1 : 0; }
int kvzx_decoder_band_finish(OpenHevc_Handle hh)
{
  Handle *h = H(hh);
  if (!h) return -1;
  h->have_pic = false;
  const int rc = h->dec->band_finish();
  if (rc > 0) h->have_pic = h->dec->get_picture(&h->pic);
  return rc;
}
void kvzx_decoder_set_output_hold(OpenHevc_Handle hh, int pictures) { Handle *h = H(hh); if (h) h->dec->set_output_hold(pictures); }
void kvzx_decoder_set_download(OpenHevc_Handle hh, int on) { Handle *h = H(hh); if (h) h->dec->set_download(on != 0); }
void kvzx_decoder_set_profiling(OpenHevc_Handle hh, int every) { Handle *h = H(hh); if (h) h->dec->set_profiling(every); }
int kvzx_decoder_kernel_times(OpenHevc_Handle hh, double *ms, uint64_t *launches, int reset)
{
  Handle *h = H(hh);
  if (!h) return 0;
  double m[kvzx::DK_COUNT]; uint64_t n[kvzx::DK_COUNT];
  h->dec->get_kernel_times(m, n, reset != 0);
  for (int i = 0; i < kvzx::DK_COUNT; i++) { if (ms) ms[i] = m[i]; if (launches) launches[i] = n[i]; }
  return kvzx::DK_COUNT;
}
int kvzx_batch_stats(int device, uint64_t *batches, uint64_t *pictures, uint64_t *sizes, double *ms, uint64_t *launches, uint64_t *frames, int reset)
{
  kvzx::BatchStats st;
  kvzx::DecBatcher::get(device).get_stats(&st, reset != 0);
  if (batches) *batches = st.batches;
  if (pictures) *pictures = st.pictures;
  if (sizes) for (int i = 0; i <= KVZ_DEC_BATCH_MAX; i++) sizes[i] = st.by_size[i];
  for (int i = 0; i < kvzx::BK_COUNT; i++) { if (ms) ms[i] = st.ms[i]; if (launches) launches[i] = st.launches[i]; if (frames) frames[i] = st.frames[i]; }
  return kvzx::BK_COUNT;
}
const char *kvzx_batch_kernel_name(int id)
{
  static const char *names[kvzx::BK_COUNT] = {"k_dec_inter_n", "k_dec_intra_n", "k_dec_deblock_n", "k_dec_sao_n"};
  return (id >= 0 && id < kvzx::BK_COUNT) ? names[id] : nullptr;
}
void kvzx_batch_hold(int device, int on) { kvzx::DecBatcher::get(device).hold(on != 0); }
const char *kvzx_decoder_kernel_name(int id)
{
  static const char *names[kvzx::DK_COUNT] = {"k_dec_inter", "k_dec_intra", "k_dec_deblock", "host_cabac_parse", "k_dec_sao", "k_dec_intra<P>"};
  return (id >= 0 && id < kvzx::DK_COUNT) ? names[id] : nullptr;
}
int kvzx_decoder_debug_copy(OpenHevc_Handle hh, const char *what, void *dst, size_t bytes)
{
  Handle *h = H(hh);
  return h && what && h->dec->debug_copy(what, dst, bytes) ? 1 : 0;
}

}  // extern "C"
