// kvazzup_amd/csrc/kvz_api.hip -- the kvz_api function table (include/kvazaar.h) and the encoder
// extensions (include/kvazzup_amd.h) on top of kvzx::Encoder.
// Drop-in for the calls uvgComm makes at /root/reference/src/media/processing/kvazaarfilter.cpp:
// 145-299 (configuration), 407-449 (encode loop), 456-476 (chunk / recon hand-back), 313-329 (close).
#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "../../include/kvazzup_amd.h"
#include <deque>
#include <mutex>
#include <unordered_set>
#include "encoder.h"

using kvzx::Encoder;
using kvzx::EncoderConfig;
using kvzx::EncodedPicture;

struct kvz_encoder {
  Encoder *impl;
  kvz_config cfg;
  uint64_t last_bins;
  int warned_rc;
  std::deque<kvz_picture *> *in_flight;      // source pictures whose output has not been returned yet (owf >= 1)
  std::deque<kvz_picture *> *in_flight_recon; // their reconstruction pictures (pic_out), allocated when the source went in: the encoder copies into them behind the picture's kernels (Encoder::set_recon_sink); NULL entries: none wanted
};

namespace {

// ----------------------------------------------------------------------------- config
kvz_config *config_alloc(void) { return (kvz_config *)calloc(1, sizeof(kvz_config)); }
int config_destroy(kvz_config *cfg) { free(cfg); return 1; }

int config_init(kvz_config *cfg)
{
  if (!cfg) return 0;
  memset(cfg, 0, sizeof(*cfg));
  cfg->qp = 22; cfg->intra_period = 64; cfg->vps_period = 0;
  cfg->framerate = 25.0; cfg->framerate_num = 25; cfg->framerate_denom = 1;
  cfg->deblock_enable = 1; cfg->sao_type = KVZ_SAO_OFF;
  cfg->ime_algorithm = KVZ_IME_HEXBS; cfg->fme_level = 0; cfg->ref_frames = 1;
  cfg->tiles_width_count = 1; cfg->tiles_height_count = 1;
  cfg->wpp = 1; cfg->owf = 0; cfg->max_merge = 5; cfg->early_skip = 1;
  cfg->gop_lowdelay = 1; cfg->gop_len = 0; cfg->gop_lp_ref_depth = 1; cfg->gop_lp_temporal_layers = 1;
  cfg->mv_constraint = KVZ_MV_CONSTRAIN_NONE; cfg->hash = KVZ_HASH_NONE;
  cfg->pu_depth_inter_min = 1; cfg->pu_depth_inter_max = 2; cfg->pu_depth_intra_min = 1; cfg->pu_depth_intra_max = 3;
  cfg->me_range = 16; cfg->gpu_device = 0; cfg->recon_output = 1;
  cfg->threads = -1;                                    // auto, as in Kvazaar
  cfg->me_early_termination = 1;                        // on, as in Kvazaar
  cfg->intra_satd = 1; cfg->gpu_entropy = 0; cfg->intra_chain = 1;
  return 1;
}

bool parse_bool(const char *v, int *out)
{
  if (!strcmp(v, "1") || !strcmp(v, "true") || !strcmp(v, "yes") || !strcmp(v, "on")) { *out = 1; return true; }
  if (!strcmp(v, "0") || !strcmp(v, "false") || !strcmp(v, "no") || !strcmp(v, "off")) { *out = 0; return true; }
  return false;
}
bool parse_int(const char *v, int *out)
{
  char *end = nullptr; errno = 0; long x = strtol(v, &end, 10);
  if (end == v || *end || errno == ERANGE || x < INT_MIN || x > INT_MAX) return false;      // (a number that does not fit is not a value of any option)
  *out = (int)x; return true;
}

int config_parse(kvz_config *cfg, const char *name, const char *value)
{
  if (!cfg || !name) return 0;
  if (!value) value = "true";
  std::string n(name);
  int iv = 0;
#define INT_OPT(key, field, lo, hi) if (n == key) { if (!parse_int(value, &iv) || iv < (lo) || iv > (hi)) return 0; cfg->field = iv; return 1; }
#define BOOL_OPT(key, field) if (n == key) { if (!parse_bool(value, &iv)) { cfg->field = 0; return *value ? 0 : 1; } cfg->field = iv; return 1; }
  if (n == "preset") {
    static const char *presets[] = {"ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo"};
    // One GPU tool set serves every preset; what the presets above ultrafast add from it is SAO and the fractional-sample
    // motion refinement (Kvazaar's preset table, as recalled in SURVEY.md appendix A: sao off and subme 0 at ultrafast only;
    // subme 2 at superfast / veryfast, 4 from faster on).  Later options ("sao", "subme") override, as in Kvazaar.
    // rdoq from medium on, signhide from slow on (as recalled from Kvazaar's table, which ties both to its slower presets)
    // intra-in-p: Kvazaar codes intra units in P pictures at every preset; here from superfast on -- the intra units of a P picture are a
    // dependency chain on the GPU (k_intra_recon<.., P>: measured 2.8x fewer pictures/s on the benchmark clip for 13 % fewer bits, DESIGN.md
    // section 2), so the fastest preset keeps the all-inter P pictures the benchmark was defined with; "intra-in-p=1" switches it on there too.
    // Two levels (round 4): 1 = 16x16 intra units only (superfast .. fast: a CTU of 8x8 intra units is twice the wavefront steps of one of 16x16 units in the
    // encoder's and the decoder's chain), 2 = 16x16 and 8x8 units (medium and slower)
    for (int i = 0; i < 10; i++) if (!strcmp(value, presets[i])) {
      cfg->sao_type = i ? KVZ_SAO_FULL : KVZ_SAO_OFF;
      cfg->fme_level = i == 0 ? 0 : (i <= 2 ? 2 : 4);
      cfg->rdoq_enable = i >= 5; cfg->signhide_enable = i >= 6; cfg->intra_in_p = i >= 5 ? 2 : (i >= 1 ? 1 : 0);
      // me-source (round 6, "uvgx search pipelining v1"): the integer search on the previous INPUT picture -- a P picture's search then runs beside the previous
      // picture's reconstruction loop instead of behind it.  On where a fractional refinement against the reconstruction follows and nothing slower is asked
      // for (superfast .. fast); measured on the checker: within +-0.6 % bits and 0.02 dB there, +1.1 .. 1.4 % bits / -0.17 dB without the refinement (ultrafast)
      cfg->me_source = i >= 1 && i <= 4;
      return 1;
    }
    return 0;
  }
  if (n == "input-res") {
    // "<w>x<h>", both within what encoder_open takes (16 384: hevc sizes here are 16-bit quantities, filter.h:61-62 VideoInfo)
    char *e1 = nullptr, *e2 = nullptr; errno = 0;
    const long w = strtol(value, &e1, 10);
    if (e1 == value || *e1 != 'x' || errno == ERANGE) return 0;
    const long h = strtol(e1 + 1, &e2, 10);
    if (e2 == e1 + 1 || *e2 || errno == ERANGE || w <= 0 || h <= 0 || w > 16384 || h > 16384) return 0;
    cfg->width = (int)w; cfg->height = (int)h; return 1;
  }
  if (n == "input-fps") {
    // "<num>/<den>" (kvazaarfilter.cpp:174) or a decimal number; either way a rate between 0 and a million pictures per second
    char *e1 = nullptr, *e2 = nullptr; errno = 0;
    const long a = strtol(value, &e1, 10);
    if (e1 != value && *e1 == '/') {
      const long b = strtol(e1 + 1, &e2, 10);
      if (e2 == e1 + 1 || *e2 || errno == ERANGE || a <= 0 || b <= 0 || a > 1000000000L || b > 1000000000L || a / b > 1000000) return 0;
      cfg->framerate_num = (int)a; cfg->framerate_denom = (int)b; cfg->framerate = (double)a / (double)b; return 1;
    }
    char *e3 = nullptr;
    const double f = strtod(value, &e3);
    if (e3 == value || *e3 || !(f > 0) || !(f <= 1000000.0)) return 0;
    cfg->framerate = f; cfg->framerate_num = (int)(f * 1000 + 0.5); cfg->framerate_denom = 1000;
    if (cfg->framerate_num % 1000 == 0) { cfg->framerate_num /= 1000; cfg->framerate_denom = 1; }
    return 1;
  }
  if (n == "threads") { if (!strcmp(value, "auto")) { cfg->threads = -1; return 1; } INT_OPT("threads", threads, -1, 4096) }
  INT_OPT("owf", owf, 0, 64)
  BOOL_OPT("wpp", wpp)
  if (n == "tiles") {
    int a = 0, b = 0;
    if (sscanf(value, "%dx%d", &a, &b) != 2 || a < 1 || b < 1) return 0;
    // kvazaar tiles=CxR: C columns x R rows, uniform spacing (at most 20 x 22, checked against the picture size in encoder_open)
    if (a > 20 || b > 22) return 0;
    cfg->tiles_width_count = a; cfg->tiles_height_count = b;
    return 1;
  }
  if (n == "slices") {
    if (!strcmp(value, "wpp")) { cfg->slices = KVZ_SLICES_WPP; return 1; }
    if (!strcmp(value, "tiles")) { cfg->slices = KVZ_SLICES_TILES; return 1; }
    if (!strcmp(value, "none") || !strcmp(value, "0")) { cfg->slices = KVZ_SLICES_NONE; return 1; }
    return 0;
  }
  INT_OPT("qp", qp, 0, 51)
  INT_OPT("period", intra_period, 0, 1 << 30)
  INT_OPT("vps-period", vps_period, 0, 1 << 30)
  INT_OPT("bitrate", target_bitrate, 0, 1 << 30)
  if (n == "rc-algorithm") {
    if (!strcmp(value, "lambda")) { cfg->rc_algorithm = KVZ_LAMBDA; return 1; }
    if (!strcmp(value, "oba")) { cfg->rc_algorithm = KVZ_OBA; return 1; }
    if (!strcmp(value, "no-rc") || !strcmp(value, "0")) { cfg->rc_algorithm = KVZ_NO_RC; return 1; }
    return 0;
  }
  BOOL_OPT("intra-bits", intra_bits)
  if (n == "gop") {
    int g = 0, d = 0, t = 0;
    if (sscanf(value, "lp-g%dd%dt%d", &g, &d, &t) == 3 && g >= 1 && g <= KVZ_MAX_GOP_LENGTH && d >= 1 && t >= 1) {
      cfg->gop_lowdelay = 1; cfg->gop_len = g; cfg->gop_lp_ref_depth = d; cfg->gop_lp_temporal_layers = t; return 1;
    }
    if (!strcmp(value, "0")) { cfg->gop_len = 0; cfg->gop_lowdelay = 1; return 1; }
    return 0;                                      // hierarchical-B GOPs are not implemented
  }
  if (n == "scaling-list") {
    if (!strcmp(value, "off") || !strcmp(value, "0")) { cfg->scaling_list = KVZ_SCALING_LIST_OFF; return 1; }
    if (!strcmp(value, "default")) { cfg->scaling_list = KVZ_SCALING_LIST_DEFAULT; return 1; }      // scaling_list_enabled_flag with the default lists (Tables 7-5 / 7-6): quantiser and dequantiser per position
    return 0;                                      // "custom" (a cqmfile) is not implemented
  }
  if (n == "mv-constraint") {
    if (!*value || !strcmp(value, "none")) { cfg->mv_constraint = KVZ_MV_CONSTRAIN_NONE; return 1; }
    if (!strcmp(value, "frame")) { cfg->mv_constraint = KVZ_MV_CONSTRAIN_FRAME; return 1; }
    if (!strcmp(value, "tile")) { cfg->mv_constraint = KVZ_MV_CONSTRAIN_TILE; return 1; }
    if (!strcmp(value, "frametile")) { cfg->mv_constraint = KVZ_MV_CONSTRAIN_FRAME_AND_TILE; return 1; }
    if (!strcmp(value, "frametilemargin")) { cfg->mv_constraint = KVZ_MV_CONSTRAIN_FRAME_AND_TILE_MARGIN; return 1; }
    return 0;
  }
  INT_OPT("vaq", vaq, 0, 20)
  if (n == "deblock") {
    int b = 0, t = 0;
    if (sscanf(value, "%d:%d", &b, &t) == 2) { cfg->deblock_enable = 1; cfg->deblock_beta = b; cfg->deblock_tc = t; return (b == 0 && t == 0) ? 1 : 0; }
    if (parse_bool(value, &iv)) { cfg->deblock_enable = iv; return 1; }
    return 0;
  }
  if (n == "sao") {
    if (!strcmp(value, "off") || !strcmp(value, "0") || !strcmp(value, "false")) { cfg->sao_type = KVZ_SAO_OFF; return 1; }
    if (!strcmp(value, "full") || !strcmp(value, "1") || !strcmp(value, "true")) { cfg->sao_type = KVZ_SAO_FULL; return 1; }
    return 0;                                           // "edge" / "band" alone: not implemented
  }
  if (n == "me") {
    static const char *names[] = {"hexbs", "tz", "full", "full8", "full16", "full32", "full64", "dia"};
    for (int i = 0; i < 8; i++) if (!strcmp(value, names[i])) { cfg->ime_algorithm = (kvz_ime_algorithm)i; return 1; }
    return 0;
  }
  INT_OPT("subme", fme_level, 0, 4)
  INT_OPT("rd", rdo, 0, 1)                       // (0 / 1: decisions by SAD / SATD as here; the full-RDO levels 2 and 3 are rejected)
  INT_OPT("ref", ref_frames, 1, 1)               // (one reference picture)
  INT_OPT("max-merge", max_merge, 1, 5)
  INT_OPT("me-steps", me_max_steps, -1, 1 << 20)
  INT_OPT("fast-residual-cost", fast_residual_cost_limit, 0, 51)
  INT_OPT("me-range", me_range, 1, 32)
  INT_OPT("gpu", gpu_device, 0, 64)
  BOOL_OPT("recon-output", recon_output)
  BOOL_OPT("intra-chain", intra_chain) BOOL_OPT("me-source", me_source) BOOL_OPT("input-hold", input_hold) INT_OPT("intra-in-p", intra_in_p, 0, 2)
  if (n == "null-input") {
    if (!strcmp(value, "drain")) { cfg->null_input_poll = 0; return 1; }
    if (!strcmp(value, "poll")) { cfg->null_input_poll = 1; return 1; }
    return 0;
  }
  INT_OPT("band-row0", band_row0, 0, 4096)
  INT_OPT("band-rows", band_rows, 0, 4096)
  BOOL_OPT("rdoq", rdoq_enable) BOOL_OPT("signhide", signhide_enable) BOOL_OPT("lossless", lossless)
  // Tools this encoder does not have: switching one ON is rejected -- config_parse's return value is all uvgComm's custom-parameter list
  // reports back (kvazaarfilter.cpp:363-367) --, switching it off is accepted.
#define OFF_ONLY(key, field) if (n == key) { if (!parse_bool(value, &iv)) return 0; cfg->field = 0; return iv ? 0 : 1; }
  OFF_ONLY("smp", smp_enable) OFF_ONLY("amp", amp_enable) OFF_ONLY("bipred", bipred) OFF_ONLY("tmvp", tmvp_enable) OFF_ONLY("transform-skip", trskip_enable)
  OFF_ONLY("full-intra-search", full_intra_search) OFF_ONLY("mv-rdo", mv_rdo) OFF_ONLY("implicit-rdpcm", implicit_rdpcm) OFF_ONLY("intra-rdo-et", intra_rdo_et)
#undef OFF_ONLY
  BOOL_OPT("rdoq-skip", rdoq_skip) BOOL_OPT("early-skip", early_skip)       // (recorded: the zero-out of "uvgx RDOQ v1" and the skip decision do not depend on them)
  BOOL_OPT("set-qp-in-cu", set_qp_in_cu) BOOL_OPT("psnr", calc_psnr) BOOL_OPT("cpuid", cpuid)
  if (n == "cu-split-termination") { cfg->cu_split_termination = !strcmp(value, "off"); return (!strcmp(value, "zero") || !strcmp(value, "off")) ? 1 : 0; }
  if (n == "intra-satd") return parse_bool(value, &cfg->intra_satd);
  if (n == "gpu-entropy") return parse_bool(value, &cfg->gpu_entropy);
  if (n == "me-early-termination") {
    if (!strcmp(value, "off")) cfg->me_early_termination = 0; else if (!strcmp(value, "on")) cfg->me_early_termination = 1;
    else if (!strcmp(value, "sensitive")) cfg->me_early_termination = 2; else return 0;
    return 1;
  }
  if (n == "hash") {
    if (!strcmp(value, "none")) { cfg->hash = KVZ_HASH_NONE; return 1; }
    if (!strcmp(value, "checksum")) { cfg->hash = KVZ_HASH_CHECKSUM; return 1; }
    if (!strcmp(value, "md5")) { cfg->hash = KVZ_HASH_MD5; return 1; }
    return 0;
  }
  if (n == "pu-depth-inter" || n == "pu-depth-intra") {
    int a = 0, b = 0;
    if (sscanf(value, "%d-%d", &a, &b) != 2 || a < 0 || b > 4 || a > b) return 0;
    if (n == "pu-depth-inter") { cfg->pu_depth_inter_min = a; cfg->pu_depth_inter_max = b; }
    else { cfg->pu_depth_intra_min = a; cfg->pu_depth_intra_max = b; }
    return 1;
  }
  if (n == "info" || n == "aud" || n == "open-gop") return parse_bool(value, &iv) ? 1 : 0;
#undef INT_OPT
#undef BOOL_OPT
  return 0;     // unknown option (kvazaarfilter.cpp:363-367 logs these)
}

// ----------------------------------------------------------------------------- pictures, chunks
// Pictures are page-locked when a HIP device is there: uvgComm copies every camera frame into a kvz_picture from picture_alloc
// (kvazaarfilter.cpp:410-418) and hands that to encoder_encode, so the copy engine can read the caller's picture where it lies --
// no second host copy into a staging buffer.  Without a device (header-only use, the CPU test suite) they are ordinary memory.
// Freed page-locked pictures are kept (up to 16) for the next picture_alloc of the same size: encoder_encode hands out a fresh reconstruction
// picture per access unit and uvgComm frees it at once (kvazaarfilter.cpp:476) -- page-locking memory costs milliseconds, a list look-up nothing.
struct PinnedSet { std::mutex m; std::unordered_set<const void *> s; std::vector<std::pair<size_t, void *>> spare; };
PinnedSet &pinned_set() { static PinnedSet *p = new PinnedSet(); return *p; }      // (never destroyed: pictures may be freed during exit)
bool is_pinned(const void *buf) { PinnedSet &ps = pinned_set(); std::lock_guard<std::mutex> l(ps.m); return ps.s.count(buf) != 0; }

kvz_picture *picture_alloc_csp(enum kvz_chroma_format csp, int32_t width, int32_t height)
{
  if (csp != KVZ_CSP_420 || width <= 0 || height <= 0 || (width & 1) || (height & 1)) return nullptr;
  kvz_picture *p = (kvz_picture *)calloc(1, sizeof(kvz_picture));
  if (!p) return nullptr;
  size_t ny = (size_t)width * height;
  void *pin = nullptr;
  int ndev = 0;
  {
    PinnedSet &ps = pinned_set(); std::lock_guard<std::mutex> l(ps.m);
    for (size_t i = 0; i < ps.spare.size(); i++) if (ps.spare[i].first == ny) { pin = ps.spare[i].second; ps.spare.erase(ps.spare.begin() + (long)i); break; }
  }
  if (pin || (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipHostMalloc(&pin, ny * 3 / 2 + 64, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && pin)) {
    p->fulldata_buf = (kvz_pixel *)pin;
    PinnedSet &ps = pinned_set(); std::lock_guard<std::mutex> l(ps.m); ps.s.insert(pin);
  } else {
    (void)hipGetLastError();
    p->fulldata_buf = (kvz_pixel *)malloc(ny * 3 / 2 + 64);
  }
  if (!p->fulldata_buf) { free(p); return nullptr; }
  p->fulldata = p->fulldata_buf;
  p->y = p->data[0] = p->fulldata; p->u = p->data[1] = p->fulldata + ny; p->v = p->data[2] = p->fulldata + ny + ny / 4;
  p->width = width; p->height = height; p->stride = width;
  p->refcount = 1; p->chroma_format = KVZ_CSP_420; p->interlacing = KVZ_INTERLACING_NONE;
  return p;
}
kvz_picture *picture_alloc(int32_t width, int32_t height) { return picture_alloc_csp(KVZ_CSP_420, width, height); }
void picture_free(kvz_picture *pic)
{
  if (!pic) return;
  if (--pic->refcount > 0) return;
  bool pinned;
  void *evicted = nullptr;
  {
    PinnedSet &ps = pinned_set(); std::lock_guard<std::mutex> l(ps.m);
    pinned = ps.s.erase(pic->fulldata_buf) != 0;
    if (pinned) {
      // newest last; the OLDEST spare goes when there are sixteen (round 5: a full list of 1080p spares left by earlier encoders kept every 4K picture of a
      // later one out -- each reconstruction picture was page-locked and released again, 3 ms apiece: 4K at uvgComm's defaults 421 -> 235 frames/s)
      ps.spare.emplace_back((size_t)pic->width * pic->height, pic->fulldata_buf);
      if (ps.spare.size() > 16) { evicted = ps.spare.front().second; ps.spare.erase(ps.spare.begin()); }
    }
  }
  if (pinned) { if (evicted) hipHostFree(evicted); } else free(pic->fulldata_buf);
  free(pic);      // roi.roi_array belongs to the caller (kvazaarfilter.cpp:53,459-463)
}
void chunk_free(kvz_data_chunk *chunk)
{
  while (chunk) { kvz_data_chunk *n = chunk->next; free(chunk); chunk = n; }
}
kvz_data_chunk *make_chunks(const uint8_t *data, size_t len)
{
  kvz_data_chunk *head = nullptr, *tail = nullptr;
  for (size_t pos = 0; pos < len || !head; pos += KVZ_DATA_CHUNK_SIZE) {
    kvz_data_chunk *c = (kvz_data_chunk *)malloc(sizeof(kvz_data_chunk));
    if (!c) { chunk_free(head); return nullptr; }
    size_t n = len - pos < KVZ_DATA_CHUNK_SIZE ? len - pos : KVZ_DATA_CHUNK_SIZE;
    memcpy(c->data, data + pos, n); c->len = (uint32_t)n; c->next = nullptr;
    if (tail) tail->next = c; else head = c;
    tail = c;
    if (len == 0) break;
  }
  return head;
}

// ----------------------------------------------------------------------------- encoder
kvz_encoder *encoder_open(const kvz_config *cfg)
{
  if (!cfg) return nullptr;
  EncoderConfig ec;
  ec.width = cfg->width; ec.height = cfg->height; ec.qp = cfg->qp; ec.intra_period = cfg->intra_period; ec.vps_period = cfg->vps_period;
  ec.me_range = cfg->me_range; ec.fps_num = cfg->framerate_num; ec.fps_den = cfg->framerate_denom;
  ec.wpp = cfg->wpp ? 1 : 0; ec.deblock = cfg->deblock_enable ? 1 : 0; ec.device = cfg->gpu_device; ec.owf = cfg->owf > 16 ? 16 : cfg->owf;
  ec.tile_rows = cfg->tiles_height_count > 1 ? cfg->tiles_height_count : 1;
  ec.tile_cols = cfg->tiles_width_count > 1 ? cfg->tiles_width_count : 1;
  // a grid finer than the CTU grid (uvgComm's "16x16" at 1080p: 17 CTU rows) is coded with as many tiles as there are CTUs in that direction
  if (ec.tile_rows > (cfg->height + 63) / 64) ec.tile_rows = (cfg->height + 63) / 64;
  if (ec.tile_cols > (cfg->width + 63) / 64) ec.tile_cols = (cfg->width + 63) / 64;
  ec.band_row0 = cfg->band_row0; ec.band_rows = cfg->band_rows;
  // "threads" (uvgComm video/kvzThreads: auto = core count, Main = 0): what is threaded on the host here is the arithmetic coder
  ec.entropy_threads = cfg->threads < 0 ? 16 : (cfg->threads == 0 ? 1 : (cfg->threads > 16 ? 16 : cfg->threads));
  ec.me_early = cfg->me_early_termination != 0;
  ec.satd = cfg->intra_satd != 0;
  ec.subme = cfg->fme_level < 0 ? 0 : (cfg->fme_level > 4 ? 4 : cfg->fme_level);
  ec.entropy_gpu = cfg->gpu_entropy != 0;
  ec.input_hold = cfg->input_hold != 0;
  ec.scaling_list = cfg->scaling_list == KVZ_SCALING_LIST_DEFAULT; ec.intra_chain = cfg->intra_chain != 0;
  // kvz_config.lossless (uvgComm writes the field itself, kvazaarfilter.cpp:244): cu_transquant_bypass in every coding unit (round 4, second half)
  ec.lossless = cfg->lossless != 0;
  ec.rdoq = cfg->rdoq_enable != 0; ec.signhide = cfg->signhide_enable != 0; ec.intra_in_p = cfg->intra_in_p; ec.me_source = cfg->me_source != 0;
  ec.hash = cfg->hash == KVZ_HASH_MD5 ? 2 : (cfg->hash == KVZ_HASH_CHECKSUM ? 1 : 0);
  ec.vaq = cfg->vaq > 0 ? cfg->vaq : 0;
  ec.qp_in_cu = (cfg->set_qp_in_cu || ec.vaq > 0) ? 1 : 0;
  ec.sao = cfg->sao_type == KVZ_SAO_FULL;
  ec.mv_frame = cfg->mv_constraint == KVZ_MV_CONSTRAIN_FRAME || cfg->mv_constraint == KVZ_MV_CONSTRAIN_FRAME_AND_TILE ? 1 : (cfg->mv_constraint == KVZ_MV_CONSTRAIN_FRAME_AND_TILE_MARGIN ? 2 : 0);   // (tile rows always confine vectors to the tile)
  ec.bitrate = cfg->target_bitrate > 0 ? cfg->target_bitrate : 0;
  // rc-algorithm lambda / oba (uvgComm sets "lambda" with its bitrate, kvazaarfilter.cpp:225-228): the picture-level controller plus feedback
  // inside the picture (rate control v2, four groups of CTU rows); left unset: the picture-level controller alone
  ec.slices = cfg->slices == KVZ_SLICES_WPP ? 1 : (cfg->slices == KVZ_SLICES_TILES ? 2 : 0);
  ec.rc_bands = (ec.bitrate > 0 && cfg->rc_algorithm != KVZ_NO_RC) ? 4 : 0;
  std::string err;
  Encoder *impl = Encoder::create(ec, &err);
  if (!impl) { fprintf(stderr, "kvazzup_amd: encoder_open failed: %s\n", err.c_str()); return nullptr; }
  kvz_encoder *e = new kvz_encoder();
  e->impl = impl; e->cfg = *cfg; e->last_bins = 0; e->warned_rc = 0; e->in_flight = new std::deque<kvz_picture *>(); e->in_flight_recon = new std::deque<kvz_picture *>();
  return e;
}
void encoder_close(kvz_encoder *e)
{
  if (!e) return;
  delete e->impl;
  for (kvz_picture *p : *e->in_flight) picture_free(p);
  delete e->in_flight;
  for (kvz_picture *p : *e->in_flight_recon) picture_free(p);
  delete e->in_flight_recon;
  delete e;
}

void fill_info(kvz_encoder *e, const EncodedPicture &ep, kvz_frame_info *info)
{
  if (!info) return;
  memset(info, 0, sizeof(*info));
  info->poc = ep.poc; info->qp = (int8_t)ep.qp;
  info->nal_unit_type = ep.is_intra ? KVZ_NAL_IDR_W_RADL : KVZ_NAL_TRAIL_R;
  info->slice_type = ep.is_intra ? KVZ_SLICE_I : KVZ_SLICE_P;
  if (!ep.is_intra) { info->ref_list_len[0] = 1; info->ref_list[0][0] = ep.poc - 1; }
}

int encoder_headers(kvz_encoder *e, kvz_data_chunk **data_out, uint32_t *len_out)
{
  if (!e || !data_out) return 0;
  kvzx::StreamParams sp;
  const EncoderConfig &c = e->impl->config();
  sp.cw = e->impl->coded_width(); sp.ch = e->impl->coded_height(); sp.width = c.width; sp.height = c.height; sp.qp = c.qp;
  sp.wpp = c.wpp; sp.deblock = c.deblock; sp.fps_num = c.fps_num; sp.fps_den = c.fps_den; sp.tile_rows = c.tile_rows; sp.tile_cols = c.tile_cols; sp.qp_in_cu = c.qp_in_cu;
  sp.sao = c.sao; sp.slices = c.slices; sp.signhide = c.signhide;
  std::vector<uint8_t> out;
  kvzx::BitWriter a, b, d;
  kvzx::write_vps(a, sp); kvzx::append_nal(out, 32, a.data().data(), a.data().size());
  kvzx::write_sps(b, sp); kvzx::append_nal(out, 33, b.data().data(), b.data().size());
  kvzx::write_pps(d, sp); kvzx::append_nal(out, 34, d.data().data(), d.data().size());
  *data_out = make_chunks(out.data(), out.size());
  if (len_out) *len_out = (uint32_t)out.size();
  return *data_out ? 1 : 0;
}

// With owf == 0 the output belongs to pic_in.  With owf >= 1 (kvazaarfilter.cpp:193) the output is the
// picture of the previous call (len_out == 0 on the first one), and a call with pic_in == NULL returns the
// picture still in flight -- the loop in KvazaarFilter::feedInput (kvazaarfilter.cpp:440-448) relies on that.
int encoder_encode(kvz_encoder *e, kvz_picture *pic_in, kvz_data_chunk **data_out, uint32_t *len_out,
                   kvz_picture **pic_out, kvz_picture **src_out, kvz_frame_info *info_out)
{
  if (data_out) *data_out = nullptr;
  if (len_out) *len_out = 0;
  if (pic_out) *pic_out = nullptr;
  if (src_out) *src_out = nullptr;
  if (!e) return 0;
  EncodedPicture ep;
  const long c0 = e->impl->collected_count(), a0 = e->impl->accepted_count();
  // a picture whose turn came and whose collection failed (its slot reported an error) is gone: its source and reconstruction pictures leave the queues with it,
  // or every later output would be paired with the picture before its own
  auto drop_failed = [&](long before) {
    if (e->impl->collected_count() == before || e->in_flight->empty()) return;
    kvz_picture *s0 = e->in_flight->front(); e->in_flight->pop_front();
    kvz_picture *r0 = e->in_flight_recon->front(); e->in_flight_recon->pop_front();
    picture_free(s0); picture_free(r0);                    // (the slot has been finished, successfully or not: nothing is queued into r0 any more)
  };
  if (!pic_in) {
    // NULL input: Kvazaar waits for the oldest picture in flight and returns it.  uvgComm calls this in a loop after EVERY picture that
    // produced output (kvazaarfilter.cpp:440-448), which with that meaning empties the pipeline each time: video/OWF pictures go in
    // back to back, then all of them are waited for.  "null-input=poll" (settable through uvgComm's custom-parameter list) makes the call
    // return only pictures that are already finished -- the loop then collects what is there and the pipeline stays full.
    if (!(e->cfg.null_input_poll ? e->impl->poll(&ep) : e->impl->flush(&ep))) { drop_failed(c0); return 0; }
  } else {
    if (pic_in->width != e->cfg.width || pic_in->height != e->cfg.height || !pic_in->y || !pic_in->u || !pic_in->v) return 0;
    // delta-QP map of this picture (kvazaarfilter.cpp:423-431); honoured when set-qp-in-cu enabled the signalling
    if (pic_in->roi.roi_array && pic_in->roi.width > 0 && pic_in->roi.height > 0) e->impl->set_roi(pic_in->roi.width, pic_in->roi.height, pic_in->roi.roi_array);
    else e->impl->set_roi(0, 0, nullptr);
    // the picture's reconstruction (pic_out, kvazaarfilter.cpp:435-448,476): its memory is handed to the encoder WITH the source, so that it is filled
    // behind the picture's kernels, beside the entropy coding, not by a download of its own after the access unit is done
    kvz_picture *rec = nullptr;
    const bool pinned_in = pic_in->y == pic_in->fulldata_buf && is_pinned(pic_in->fulldata_buf);
    if (pic_out && e->cfg.recon_output) {
      rec = picture_alloc(e->cfg.width, e->cfg.height);
      if (rec && is_pinned(rec->fulldata_buf)) e->impl->set_recon_sink(rec->y, rec->u, rec->v);
    }
    if (!e->impl->encode_host(pic_in->y, pic_in->u, pic_in->v, &ep, pinned_in)) {
      // Two different failures.  The picture never went in (accepted count unchanged): its reconstruction picture is ours to free.  Or it DID go in and what
      // failed is the collection of the oldest picture in flight: then the copy into rec may already be queued behind this picture's kernels -- the pair
      // joins the queues like any accepted picture's (freed when its own turn comes, after the encoder has finished with the sink), and the picture whose
      // collection failed leaves them
      if (e->impl->accepted_count() == a0) { e->impl->set_recon_sink(nullptr, nullptr, nullptr); picture_free(rec); }
      else { pic_in->refcount++; e->in_flight->push_back(pic_in); e->in_flight_recon->push_back(rec); }
      drop_failed(c0);
      return 0;
    }
    pic_in->refcount++;
    e->in_flight->push_back(pic_in);
    e->in_flight_recon->push_back(rec);
  }
  if (!ep.valid) return 1;
  kvz_picture *src = e->in_flight->front();
  e->in_flight->pop_front();
  kvz_picture *rec = e->in_flight_recon->front();
  e->in_flight_recon->pop_front();
  e->last_bins = ep.bins;
  bool ok = true;
  if (data_out) { *data_out = make_chunks(ep.au.data(), ep.au.size()); ok = *data_out != nullptr; }
  if (len_out) *len_out = (uint32_t)ep.au.size();
  if (ok && pic_out && e->cfg.recon_output) {
    kvz_picture *r = rec ? rec : picture_alloc(e->cfg.width, e->cfg.height);
    rec = nullptr;
    if (!r || (!ep.recon_delivered && !e->impl->download_recon(r->y, r->u, r->v))) { picture_free(r); ok = false; }      // (not delivered: the picture was not page-locked, or came in before pic_out was asked for)
    else { r->pts = src->pts; r->dts = src->dts; *pic_out = r; }
  }
  picture_free(rec);
  if (src_out && ok) *src_out = src; else picture_free(src);
  fill_info(e, ep, info_out);
  return ok ? 1 : 0;
}

const kvz_api kApi = {
  config_alloc, config_destroy, config_init, config_parse,
  picture_alloc, picture_free, chunk_free,
  encoder_open, encoder_close, encoder_headers, encoder_encode, picture_alloc_csp,
};

}  // namespace

extern "C" {

const kvz_api *kvz_api_get(int bit_depth) { return bit_depth == 8 ? &kApi : nullptr; }

const char *kvzx_version(void) { return "kvazzup_amd 0.1 (gfx950)"; }
int kvzx_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

static int finish_raw(kvz_encoder *e, const EncodedPicture &ep, uint8_t *au_buf, uint32_t au_cap, uint32_t *len_out, kvz_frame_info *info)
{
  if (len_out) *len_out = 0;
  if (!ep.valid) return 1;                     // owf >= 1: nothing output by this call
  e->last_bins = ep.bins;
  if (len_out) *len_out = (uint32_t)ep.au.size();
  fill_info(e, ep, info);
  if (ep.au.size() > au_cap || !au_buf) return 0;
  memcpy(au_buf, ep.au.data(), ep.au.size());
  return 1;
}
int kvzx_encoder_encode_device(kvz_encoder *e, const void *d_i420, uint8_t *au_buf, uint32_t au_cap, uint32_t *len_out, kvz_frame_info *info)
{
  if (!e) return 0;
  EncodedPicture ep;
  if (!d_i420) { if (!e->impl->flush(&ep)) return 0; }          // NULL input: return the picture in flight (owf >= 1)
  else if (!e->impl->encode_device((const uint8_t *)d_i420, &ep)) return 0;
  return finish_raw(e, ep, au_buf, au_cap, len_out, info);
}
int kvzx_encoder_encode_host(kvz_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v, uint8_t *au_buf, uint32_t au_cap,
                             uint32_t *len_out, kvz_frame_info *info)
{
  if (!e || !y || !u || !v) return 0;
  EncodedPicture ep;
  if (!e->impl->encode_host(y, u, v, &ep)) return 0;
  return finish_raw(e, ep, au_buf, au_cap, len_out, info);
}
int kvzx_encoder_coded_size(kvz_encoder *e, int *cw, int *ch)
{
  if (!e) return 0;
  if (cw) *cw = e->impl->coded_width();
  if (ch) *ch = e->impl->coded_height();
  return 1;
}
int kvzx_encoder_download_recon(kvz_encoder *e, uint8_t *y, uint8_t *u, uint8_t *v) { return e && e->impl->download_recon(y, u, v) ? 1 : 0; }
int kvzx_encoder_recon_device(kvz_encoder *e, const void **planes)
{
  if (!e || !planes) return 0;
  for (int c = 0; c < 3; c++) planes[c] = e->impl->device_recon(c);
  return 1;
}
int kvzx_encoder_debug_copy(kvz_encoder *e, const char *what, void *dst, size_t bytes) { return e && what && e->impl->debug_copy(what, dst, bytes) ? 1 : 0; }
void kvzx_encoder_set_profiling(kvz_encoder *e, int every) { if (e) e->impl->set_profiling(every); }
int kvzx_encoder_kernel_times(kvz_encoder *e, double *ms, uint64_t *launches, int reset)
{
  if (!e) return 0;
  double m[kvzx::K_COUNT]; uint64_t n[kvzx::K_COUNT];
  e->impl->get_kernel_times(m, n, reset != 0);
  for (int i = 0; i < kvzx::K_COUNT; i++) { if (ms) ms[i] = m[i]; if (launches) launches[i] = n[i]; }
  return kvzx::K_COUNT;
}
const char *kvzx_encoder_kernel_name(int id)
{
  static const char *names[kvzx::K_COUNT] = {"k_pad_input", "k_me", "k_inter_recon", "k_inter_signal", "k_intra_analyse", "k_intra_recon", "k_deblock", "k_tokenize", "host_arith_coder", "k_sao", "k_tok_compact", "k_cabac_rows", "k_subpel", "k_intra_analyse<P>", "k_intra_recon<P>"};
  return (id >= 0 && id < kvzx::K_COUNT) ? names[id] : nullptr;
}
uint64_t kvzx_encoder_last_bins(kvz_encoder *e) { return e ? e->last_bins : 0; }
int kvzx_encoder_pending(kvz_encoder *e) { return e ? e->impl->pending() : 0; }
void kvzx_encoder_set_roi(kvz_encoder *e, int w, int h, const int8_t *map) { if (e) e->impl->set_roi(w, h, map); }

// ---- tile-row split of one picture over several encoders (include/kvazzup_amd.h) ----
void kvzx_encoder_band_report_au(kvz_encoder *e, long picture, uint32_t bytes) { if (e) e->impl->band_report_au(picture, bytes); }
int kvzx_encoder_band_phase1(kvz_encoder *e, const void *d_i420) { return e && e->impl->band_phase1((const uint8_t *)d_i420) ? 1 : 0; }
size_t kvzx_encoder_band_halo_bytes(kvz_encoder *e) { return e ? e->impl->halo_bytes() : 0; }
int kvzx_encoder_band_export_halo(kvz_encoder *e, void *d_up, void *d_down) { return e && e->impl->band_export_halo((uint8_t *)d_up, (uint8_t *)d_down) ? 1 : 0; }
int kvzx_encoder_band_import_halo(kvz_encoder *e, const void *d_from_up, const void *d_from_down) { return e && e->impl->band_import_halo((const uint8_t *)d_from_up, (const uint8_t *)d_from_down) ? 1 : 0; }
int kvzx_encoder_band_phase2a(kvz_encoder *e) { return e && e->impl->band_phase2a() ? 1 : 0; }
static int band_phase2_out(kvz_encoder *e, bool both, uint8_t *buf, uint32_t cap, uint32_t *sizes, int max_sub, int *nsub_out, kvz_frame_info *info);
int kvzx_encoder_band_phase2b(kvz_encoder *e, uint8_t *buf, uint32_t cap, uint32_t *sizes, int max_sub, int *nsub_out, kvz_frame_info *info) { return band_phase2_out(e, false, buf, cap, sizes, max_sub, nsub_out, info); }
int kvzx_encoder_band_phase2(kvz_encoder *e, uint8_t *buf, uint32_t cap, uint32_t *sizes, int max_sub, int *nsub_out, kvz_frame_info *info) { return band_phase2_out(e, true, buf, cap, sizes, max_sub, nsub_out, info); }
static int band_phase2_out(kvz_encoder *e, bool both, uint8_t *buf, uint32_t cap, uint32_t *sizes, int max_sub, int *nsub_out, kvz_frame_info *info)
{
  if (!e || !buf || !sizes || !nsub_out) return 0;
  std::vector<std::vector<uint8_t>> subs; EncodedPicture ep;
  if (!(both ? e->impl->band_phase2(&subs, &ep) : e->impl->band_phase2b(&subs, &ep))) return 0;
  if ((int)subs.size() > max_sub) return 0;
  size_t o = 0;
  for (size_t k = 0; k < subs.size(); k++) {
    if (o + subs[k].size() > cap) return 0;
    memcpy(buf + o, subs[k].data(), subs[k].size()); o += subs[k].size(); sizes[k] = (uint32_t)subs[k].size();
  }
  *nsub_out = (int)subs.size();
  e->last_bins = ep.bins;
  fill_info(e, ep, info);
  return 1;
}
// Host only (no GPU): the access unit from the substreams of all bands in picture order, as rank 0 does after gathering them.
// cfg: the configuration every band encoder was opened with; data: the substreams back to back, sizes[nsub] their lengths.
int kvzx_assemble_access_unit(const kvz_config *cfg, int idr, int poc, int write_parameter_sets, int slice_qp, const uint8_t *data, const uint32_t *sizes, int nsub,
                              uint8_t *out, uint32_t cap, uint32_t *len_out)
{
  if (!cfg || !data || !sizes || nsub < 1 || !out || !len_out) return 0;
  kvzx::StreamParams sp;
  sp.width = cfg->width; sp.height = cfg->height; sp.cw = (cfg->width + 63) & ~63; sp.ch = (cfg->height + 63) & ~63;
  if (sp.cw < 128) sp.cw = 128;
  sp.qp = cfg->qp; sp.wpp = cfg->wpp ? 1 : 0; sp.deblock = cfg->deblock_enable ? 1 : 0; sp.fps_num = cfg->framerate_num; sp.fps_den = cfg->framerate_denom;
  sp.tile_rows = cfg->tiles_height_count > 1 ? cfg->tiles_height_count : 1; sp.tile_cols = 1; sp.qp_in_cu = (cfg->set_qp_in_cu || cfg->vaq > 0) ? 1 : 0;
  sp.slices = (cfg->slices == KVZ_SLICES_WPP && cfg->wpp) ? 1 : ((cfg->slices == KVZ_SLICES_TILES && sp.tile_rows > 1) ? 2 : 0);
  sp.signhide = cfg->signhide_enable != 0;
  if (nsub != (sp.wpp ? sp.ch / 64 : sp.tile_rows)) return 0;
  std::vector<std::vector<uint8_t>> rows((size_t)nsub);
  size_t o = 0;
  for (int k = 0; k < nsub; k++) { rows[(size_t)k].assign(data + o, data + o + sizes[k]); o += sizes[k]; }
  std::vector<uint8_t> au;
  if (!kvzx::assemble_access_unit(au, sp, idr != 0, poc, write_parameter_sets != 0, rows, nsub, slice_qp - cfg->qp)) return 0;
  *len_out = (uint32_t)au.size();
  if (au.size() > cap) return 0;
  memcpy(out, au.data(), au.size());
  return 1;
}

}  // extern "C"
