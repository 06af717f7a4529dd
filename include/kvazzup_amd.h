/*
 * include/kvazzup_amd.h -- extensions next to the two drop-in ABIs (kvazaar.h, openHevcWrapper.h):
 * device-resident entry points used by bench.py (inputs already in HBM), per-kernel timing taken
 * with HIP events on the library's own stream, and debug read-back used by the parity tests.
 * uvgComm itself needs none of these; they exist because its boundary hands over host buffers
 * (/root/reference/src/media/processing/kvazaarfilter.cpp:410-418 memcpy into kvz_picture,
 * openhevcfilter.cpp:218-229 memcpy out of the decoder's frame).
 */
#ifndef KVAZZUP_AMD_EXT_H_
#define KVAZZUP_AMD_EXT_H_
#include "kvazaar.h"
#include "openHevcWrapper.h"
#ifdef __cplusplus
extern "C" {
#endif

KVZ_PUBLIC const char *kvzx_version(void);
/* number of HIP devices visible (0 when the runtime cannot be initialised); touches the GPU */
KVZ_PUBLIC int kvzx_device_count(void);

/* ---- encoder ---- */
/* Encode one picture given as packed I420 (width*height*3/2 bytes) in DEVICE memory of the
 * encoder's GPU.  The access unit is written to au_buf (host).  Returns 1 on success, 0 on
 * failure or when au_cap is too small (*len_out still receives the needed size). */
KVZ_PUBLIC int kvzx_encoder_encode_device(kvz_encoder *enc, const void *d_i420, uint8_t *au_buf, uint32_t au_cap,
                                          uint32_t *len_out, kvz_frame_info *info_out);
/* Same with host planes (stride = width), without going through kvz_picture / chunk lists. */
KVZ_PUBLIC int kvzx_encoder_encode_host(kvz_encoder *enc, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                                        uint8_t *au_buf, uint32_t au_cap, uint32_t *len_out, kvz_frame_info *info_out);
KVZ_PUBLIC int kvzx_encoder_coded_size(kvz_encoder *enc, int *coded_w, int *coded_h);
/* cropped reconstruction of the last coded picture into host planes (stride = width) */
KVZ_PUBLIC int kvzx_encoder_download_recon(kvz_encoder *enc, uint8_t *y, uint8_t *u, uint8_t *v);
/* device pointers (coded size, pitch = coded width [/2]) of the last reconstruction */
KVZ_PUBLIC int kvzx_encoder_recon_device(kvz_encoder *enc, const void **planes /*[3]*/);
/* debug read-back of an internal array of the last coded picture: "cu_log2", "cu_intra", "cu_flags",
 * "cu_merge_idx", "cu_mvp_idx", "cu_intra_mode", "cu_cbf" (one byte per 8x8 block), "cu_mv", "cu_mvd"
 * (two int16 per 8x8 block), "coef0".."coef2" (int16 planes), "rec0".."rec2", "src0".."src2" */
KVZ_PUBLIC int kvzx_encoder_debug_copy(kvz_encoder *enc, const char *what, void *dst, size_t bytes);
KVZ_PUBLIC void kvzx_encoder_set_profiling(kvz_encoder *enc, int every);   /* 0 off, 1 every picture, n every n-th picture */
#define KVZX_MAX_KERNELS 16
/* accumulated HIP-event time (ms) and launch count per kernel id since the last reset */
KVZ_PUBLIC int kvzx_encoder_kernel_times(kvz_encoder *enc, double *ms, uint64_t *launches, int reset);
KVZ_PUBLIC const char *kvzx_encoder_kernel_name(int id);       /* NULL past the last id */
KVZ_PUBLIC uint64_t kvzx_encoder_last_bins(kvz_encoder *enc);  /* CABAC bins of the last picture */
KVZ_PUBLIC int kvzx_encoder_pending(kvz_encoder *enc);         /* pictures handed in whose access unit has not been returned yet */

/* ---- decoder ---- */
/* Decode one NAL unit whose OUTPUT is wanted in device memory: like libOpenHevcDecode. */
/* select the HIP device ordinal; call between libOpenHevcInit and libOpenHevcStartDecoder (default 0,
 * or the environment variable KVAZZUP_AMD_DEVICE) */
KVZ_PUBLIC int kvzx_decoder_set_device(OpenHevc_Handle h, int device);
KVZ_PUBLIC int kvzx_decoder_last_error(OpenHevc_Handle h);
/* Test / measurement hook, not a decoding mode: before libOpenHevcStartDecoder.  The decoder then runs its HOST half only (NAL units, parameter sets, slice
 * headers, the CABAC slice-data parser with `parse_threads` row threads), touches no device and never outputs a picture (libOpenHevcDecode returns 0 or an
 * error).  kvzx_decoder_parse_probe_stats: out5 = pictures parsed, transform blocks, level words, FNV-1a digest of everything the parser produced, 0;
 * parse_ms = time spent in the slice-data parser. */
KVZ_PUBLIC int kvzx_decoder_set_parse_only(OpenHevc_Handle h, int parse_threads);
KVZ_PUBLIC int kvzx_decoder_parse_probe_stats(OpenHevc_Handle h, uint64_t *out5, double *parse_ms);
/* with libOpenHevcSetCheckMD5(h, 1): decoded picture hash SEI messages (MD5 / checksum, H.265 D.2.19) compared so far and how many did not
 * match the picture as decoded (each mismatch also sets the last error to -4 and is said on stderr; the picture is still handed out) */
KVZ_PUBLIC void kvzx_decoder_hash_stats(OpenHevc_Handle h, int *checked, int *mismatch);
/* device pointers (pitch = coded width [/2]) of the picture returned by the last libOpenHevcGetOutput */
KVZ_PUBLIC int kvzx_decoder_output_device(OpenHevc_Handle h, const void **planes /*[3]*/, int *pitches /*[3]*/);
/* when 0, libOpenHevcDecode leaves the picture in HBM and libOpenHevcGetOutput returns NULL planes */
KVZ_PUBLIC void kvzx_decoder_set_download(OpenHevc_Handle h, int on);
/* Lifetime of the device planes of kvzx_decoder_output_device (download off): they belong to the decoder's picture buffer and stay
 * untouched -- read as a reference picture at most -- while the next `pictures` pictures are decoded (default 2, 1..8); after that the
 * buffer may be reused.  A consumer that lags further must copy, or raise this. */
KVZ_PUBLIC void kvzx_decoder_set_output_hold(OpenHevc_Handle h, int pictures);
KVZ_PUBLIC void kvzx_decoder_set_profiling(OpenHevc_Handle h, int every);
KVZ_PUBLIC int kvzx_decoder_kernel_times(OpenHevc_Handle h, double *ms, uint64_t *launches, int reset);
KVZ_PUBLIC const char *kvzx_decoder_kernel_name(int id);
KVZ_PUBLIC int kvzx_decoder_debug_copy(OpenHevc_Handle h, const char *what, void *dst, size_t bytes);
/* ---- several decoders in one process on one device (uvgComm: one OpenHEVCFilter per peer, filtergraph.cpp:561-589): their pictures are launched by
 * the device's submission layer, same kernels of different decoders' pictures as ONE launch (csrc/batch.h).  Statistics since the last reset:
 * sizes[n] = batches of n pictures (n = 0..8, 9 entries); per kernel (kvzx_batch_kernel_name: 4 of them) the profiled batches' time, launches and
 * pictures (profiling as set with kvzx_decoder_set_profiling on any of the decoders).  Returns the number of kernels. */
KVZ_PUBLIC int kvzx_batch_stats(int device, uint64_t *batches, uint64_t *pictures, uint64_t *sizes /*[9]*/, double *ms /*[4]*/, uint64_t *launches /*[4]*/, uint64_t *frames /*[4]*/, int reset);
KVZ_PUBLIC const char *kvzx_batch_kernel_name(int id);
/* measurement aid: while held, the submission layer launches nothing and the decoders' pictures pile up; releasing launches them in full batches */
KVZ_PUBLIC void kvzx_batch_hold(int device, int on);

/* ---- tile-row split of ONE picture over several encoders, one per GPU / process (BASELINE configs[4]).
 * Every encoder is opened with the same configuration plus "tiles" = 1xN and "band-row0" / "band-rows" (CTU rows, whole
 * tiles).  Per picture and rank: phase1 (input, decisions, reconstruction, vertical-edge deblocking of the band) ->
 * export_halo -> exchange with rank - 1 (gets `up`) and rank + 1 (gets `down`), e.g. RCCL send/recv of the two device
 * blocks -> import_halo (NULL where there is no neighbour) -> phase2 (horizontal edges incl. the band's boundary edges,
 * tokenizer, arithmetic coding): the band's substreams, back to back in buf with sizes[].  Rank 0 gathers all substreams in
 * picture order and calls kvzx_assemble_access_unit (host only).  kvazzup_amd/tilesplit.py drives this over torch.distributed. */
/* delta-QP map for the pictures submitted through the device entry point (the host entry point takes it from kvz_picture.roi,
 * kvazaarfilter.cpp:423-431): w x h int8 cells over the picture, w == 0 removes it; needs "set-qp-in-cu" = 1 */
KVZ_PUBLIC void kvzx_encoder_set_roi(kvz_encoder *enc, int w, int h, const int8_t *map);
/* Rate control ("bitrate" > 0) in band mode: every band's encoder runs the same picture-level controller, which books the size of access
 * unit t - 3 before picture t; the caller tells EVERY encoder the size of each assembled access unit (picture = 0, 1, ...) before
 * picture + 3 is started.  kvazzup_amd/tilesplit.py passes the sizes along with the substream headers it gathers anyway. */
KVZ_PUBLIC void kvzx_encoder_band_report_au(kvz_encoder *enc, long picture, uint32_t bytes);
KVZ_PUBLIC int kvzx_encoder_band_phase1(kvz_encoder *enc, const void *d_i420);
KVZ_PUBLIC size_t kvzx_encoder_band_halo_bytes(kvz_encoder *enc);
KVZ_PUBLIC int kvzx_encoder_band_export_halo(kvz_encoder *enc, void *d_up, void *d_down);
KVZ_PUBLIC int kvzx_encoder_band_import_halo(kvz_encoder *enc, const void *d_from_up, const void *d_from_down);
KVZ_PUBLIC int kvzx_encoder_band_phase2(kvz_encoder *enc, uint8_t *buf, uint32_t cap, uint32_t *sizes, int max_sub, int *nsub_out, kvz_frame_info *info);
/* The same in two halves, so that the halo exchange runs beside the first: 2a (inner horizontal edges, tokenizer, arithmetic coder) needs
 * nothing from the neighbouring bands and may be called BEFORE kvzx_encoder_band_import_halo; 2b (the band's two boundary edges) after it. */
KVZ_PUBLIC int kvzx_encoder_band_phase2a(kvz_encoder *enc);
KVZ_PUBLIC int kvzx_encoder_band_phase2b(kvz_encoder *enc, uint8_t *buf, uint32_t cap, uint32_t *sizes, int max_sub, int *nsub_out, kvz_frame_info *info);
KVZ_PUBLIC int kvzx_assemble_access_unit(const kvz_config *cfg, int idr, int poc, int write_parameter_sets, int slice_qp, const uint8_t *data,
                                          const uint32_t *sizes, int nsub, uint8_t *out, uint32_t cap, uint32_t *len_out);

/* ---- the decoding side of the tile-row split: one decoder per GPU / process, each reconstructing CTU rows [row0, row0 + nrows) (whole tile rows)
 * of a stream whose motion vectors stay inside their tiles (what the split encoder writes; no SAO, no temporal motion prediction; synchronous decoder).
 * Every decoder gets every NAL unit and parses its own rows' substreams only.  Per picture: libOpenHevcDecode (returns 0 once the band is
 * reconstructed) -> band_export(0, down) -> exchange: to rank + 1, from rank - 1 -> band_import(0, from_up) -> band_deblock -> band_export(1, up)
 * -> exchange: to rank - 1, from rank + 1 -> band_import(1, from_down) -> band_finish (> 0: the picture is the output, this band's rows of it valid).
 * Halo blocks are kvzx_decoder_band_halo_bytes() long, in device memory; the calls that have no neighbour on their side are skipped. */
KVZ_PUBLIC int kvzx_decoder_set_band(OpenHevc_Handle h, int row0, int nrows);          /* before the first picture */
KVZ_PUBLIC size_t kvzx_decoder_band_halo_bytes(OpenHevc_Handle h);                     /* after the first picture's parameter sets: 8 bytes per luma column */
KVZ_PUBLIC int kvzx_decoder_band_export(OpenHevc_Handle h, int stage, void *d_buf);
KVZ_PUBLIC int kvzx_decoder_band_import(OpenHevc_Handle h, int stage, const void *d_buf);
KVZ_PUBLIC int kvzx_decoder_band_ready(OpenHevc_Handle h);      /* 1: the band of a picture is reconstructed and waits for the exchange (after the picture's last slice segment) */
KVZ_PUBLIC int kvzx_decoder_band_deblock(OpenHevc_Handle h);
KVZ_PUBLIC int kvzx_decoder_band_finish(OpenHevc_Handle h);

/* ---- row f1: I420 -> RGB32, the conversion uvgComm runs on every decoded picture before display
 * (YUVtoRGB32::process, src/media/processing/yuvtorgb32.cpp:29-64 -> yuv420_to_rgb_i_{avx2_mt,avx2,sse41,c},
 * src/media/processing/yuvconversions.cpp:72-493).  The reference has two arithmetics; `variant` picks:
 *   0 = what the filter picks on an AVX2/SSE4.1 host: the SIMD arithmetic when width % 16 == 0, else the scalar one
 *   1 = yuv420_to_rgb_i_c (every fourth output byte left as found), 2 = the SIMD converters (fourth byte 0; width % 8 == 0)
 * Planes and output in device memory; returns 1 when the kernel was launched on `hip_stream`. */
KVZ_PUBLIC int kvzx_yuv420_to_rgb32_device(const void *d_y, const void *d_u, const void *d_v, int y_pitch, int c_pitch, void *d_rgb32,
                                            int width, int height, int variant, void *hip_stream);
/* host buffers in and out, like yuv420_to_rgb_i_*(input, output, width, height): upload, convert, download */
KVZ_PUBLIC int kvzx_yuv420_to_rgb32(const uint8_t *i420, uint8_t *rgb32, int width, int height, int variant);
/* the picture returned by the last libOpenHevcGetOutput, converted where it lies in HBM (no host copy); synchronous */
KVZ_PUBLIC int kvzx_decoder_output_rgb32_device(OpenHevc_Handle h, void *d_rgb32, int variant);

/* RGB32 -> I420, the converters of src/media/processing/yuvconversions.cpp:634-797, bit for bit (their quirks included: see
 * kvazzup_amd/csrc/color_kernels.hip).  variant 1 = rgb_to_yuv420_i_c, 2 = rgb_to_yuv420_i_sse41 (which turns the picture upside
 * down).  width % 4 == 0, height % 2 == 0.  Output: packed I420 (width * height * 3 / 2 bytes). */
KVZ_PUBLIC int kvzx_rgb32_to_yuv420_device(const void *d_rgb32, void *d_i420, int width, int height, int variant, void *hip_stream);
KVZ_PUBLIC int kvzx_rgb32_to_yuv420(const uint8_t *rgb32, uint8_t *i420, int width, int height, int variant);

/* ---- filter-graph harness (kvazzup_amd/csrc/filters.h): KvazaarFilter -> [WireAdapter -> OpenHEVCFilter] ----
 * A Qt-free restatement of the two uvgComm filters on this path, each on its own thread with the
 * reference's input-buffer contract (src/media/processing/filter.cpp:151-222,364-417), driven through the
 * two ABIs above.  settings_text: "key=value" lines, key names of src/settingskeys.h:36-67 ("video/QP",
 * "video/Intra", "video/ResolutionWidth" ...) plus "uvgx/gpu", "uvgx/decoderDownload", "parameters/<i>/Name|Value". */
KVZ_PUBLIC void *uvgx_pipeline_create(const char *settings_text, int loopback_decode, int keep_outputs);
KVZ_PUBLIC int uvgx_pipeline_push_host(void *p, const uint8_t *i420, int w, int h, int fps_num, int fps_den, int64_t pts);
KVZ_PUBLIC int uvgx_pipeline_push_device(void *p, const void *d_i420, int w, int h, int fps_num, int fps_den, int64_t pts);
KVZ_PUBLIC int uvgx_pipeline_wait(void *p, uint64_t n_outputs, int timeout_ms);
/* push_device for a source that paces itself: sleeps until the encoder filter buffers fewer than max_backlog pictures (uvgComm filters
 * drop inputs at 10 buffered, filter.cpp:151-222); 0 = timed out or rejected */
KVZ_PUBLIC int uvgx_pipeline_push_device_paced(void *p, const void *d_i420, int w, int h, int fps_num, int fps_den, int64_t pts, uint32_t max_backlog, int timeout_ms);
/* the same for a HOST picture -- the reference's own boundary: the encoder filter copies it into a kvz_picture and calls encoder_encode
 * (kvazaarfilter.cpp:410-438).  borrow != 0: no copy into the Data object, the caller keeps i420 unchanged until the picture is encoded */
KVZ_PUBLIC int uvgx_pipeline_push_host_paced(void *p, const uint8_t *i420, int w, int h, int fps_num, int fps_den, int64_t pts, uint32_t max_backlog, int timeout_ms, int borrow);
KVZ_PUBLIC uint32_t uvgx_pipeline_encoder_backlog(void *p);
/* harness only (uvgComm never flushes a running graph): the pictures held back by video/OWF and by the decoder's frame threads
 * come out without further input -- the encoder filter runs its encoder_encode(NULL) loop to the end, the wire adapter sends
 * end-of-sequence NAL units.  The next picture pushed should be an IDR. */
KVZ_PUBLIC int uvgx_pipeline_flush(void *p);
KVZ_PUBLIC int uvgx_pipeline_pop_encoded(void *p, uint8_t *buf, uint32_t cap, uint32_t *size, int64_t *pts);
KVZ_PUBLIC int uvgx_pipeline_pop_decoded(void *p, uint8_t *buf, uint32_t cap, uint32_t *size, int *w, int *h, int64_t *pts);
KVZ_PUBLIC void uvgx_pipeline_stats(void *p, uint64_t *out8);
KVZ_PUBLIC void uvgx_pipeline_busy_ms(void *p, double *out3);   /* time inside process(): encoder, wire adapter, decoder filter */
KVZ_PUBLIC void uvgx_pipeline_avg_queue(void *p, double *out3); /* inputs found buffered by an arriving input, averaged: encoder, wire adapter, decoder filter */
/* per-picture delays in microseconds (what uvgComm's statistics window shows: kvazaarfilter.cpp:478-479 encoding delay, displayfilter.cpp:113-115 total
   delay).  which 0: picture pushed -> access unit out of the encoder filter; 1: -> decoded picture out of the last filter.  Returns the samples held since
   the last reset (up to `cap` copied, in output order). */
KVZ_PUBLIC uint32_t uvgx_pipeline_latency_us(void *p, int which, uint32_t *out, uint32_t cap, int reset);
KVZ_PUBLIC void uvgx_pipeline_delay_stats(void *p, double *out8);  /* the filters' own histograms: count, mean, p50, p99 of the encoding delay; the same of the total delay */
KVZ_PUBLIC void *uvgx_pipeline_encoder(void *p);     /* kvz_encoder* of the KvazaarFilter */
KVZ_PUBLIC void *uvgx_pipeline_decoder(void *p);     /* OpenHevc_Handle of the OpenHEVCFilter (NULL without loop-back) */
KVZ_PUBLIC void uvgx_pipeline_destroy(void *p);

/* ---- measurement harness (kvazzup_amd/csrc/harness_kernels.hip): the synthetic clip of SURVEY.md 8(d) generated in device memory, the buffers to
 * keep it in, a luma error sum -- so that bench.py needs no tensor library in the process (one that ships its own HIP runtime would replace the
 * system's for this library too).  Nothing of the codec. */
KVZ_PUBLIC void *kvzx_harness_alloc(int device, size_t bytes);           /* device memory; NULL on failure */
KVZ_PUBLIC void kvzx_harness_free(void *p);
KVZ_PUBLIC int kvzx_harness_sync(int device);                            /* hipDeviceSynchronize */
KVZ_PUBLIC int kvzx_harness_download(void *host, const void *dev, size_t bytes);
KVZ_PUBLIC int kvzx_harness_upload(void *dev, const void *host, size_t bytes);
/* uvgx-synth-v1 (kind 0 moving objects, 1 flat, 2 noise; kvazzup_amd/synth.py is the statement) as packed I420 into d_i420 (w*h*3/2 bytes) */
KVZ_PUBLIC int kvzx_harness_synth_frame(void *d_i420, int kind, uint32_t seed, int w, int h, int t);
/* sum of squared differences of the luma planes of a packed picture (pitch w) and a pitched one; < 0 on failure */
KVZ_PUBLIC double kvzx_harness_luma_sse(const void *d_a, const void *d_b, int w, int h, int pitch_b);
/* HBM peak in GB/s from the device properties (2 x memory clock x bus width / 8); 0 when the runtime does not report them */
KVZ_PUBLIC double kvzx_harness_hbm_peak_gbs(int device);
KVZ_PUBLIC int kvzx_harness_device_info(int device, char *name, int name_cap, int *cus, int *clock_mhz, int *mem_clock_mhz, int *bus_bits);

#ifdef __cplusplus
}
#endif
#endif
