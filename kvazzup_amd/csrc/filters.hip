// kvazzup_amd/csrc/filters.hip -- see filters.h.  Host-only C++ (compiled with the rest of the
// library).  Each function cites the reference lines whose observable behaviour it reproduces.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>
#include "filters.h"
#include "host_pool.h"

namespace uvgx {

int64_t now_ms()
{
  return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

int64_t now_us()
{
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void DelayHist::add(int64_t us)
{
  if (us < 0) us = 0;
  int b = 0;
  if (us >= 16) {                                          // quarter octaves: 4 * log2(us / 16)
    const int lg = 63 - __builtin_clzll((unsigned long long)us);      // floor(log2 us) >= 4
    const uint64_t frac = ((uint64_t)us << 2 >> lg) & 3;               // the two bits below the leading one
    b = (lg - 4) * 4 + (int)frac;
    if (b >= kBuckets) b = kBuckets - 1;
  }
  n[b]++; count++; sumUs += (uint64_t)us;
  uint64_t m = maxUs.load(std::memory_order_relaxed);
  while ((uint64_t)us > m && !maxUs.compare_exchange_weak(m, (uint64_t)us, std::memory_order_relaxed)) {}
}
double DelayHist::percentile(double q) const
{
  const uint64_t total = count.load();
  if (!total) return 0.0;
  const uint64_t want = (uint64_t)(q * (double)(total - 1)) + 1;
  uint64_t acc = 0;
  for (int b = 0; b < kBuckets; b++) {
    acc += n[b].load();
    if (acc >= want) { const int lg = b / 4 + 4, fr = b % 4; return (double)(1ull << lg) * (1.0 + (fr + 1) / 4.0); }
  }
  return (double)maxUs.load();
}

// ----------------------------------------------------------------------------------------------- Filter
Filter::Filter(std::string id, std::string name, Stats *stats, DataType input, DataType output)
    : id_(std::move(id)), name_(std::move(name)), stats_(stats), input_(input), output_(output) {}

Filter::~Filter() { stop(); }

void Filter::start()
{
  if (threadRunning_) return;
  running_ = true;
  threadRunning_ = true;
  thread_ = std::thread([this] { kvzx::name_this_thread(("kvzx-" + name_).substr(0, 15).c_str()); run(); threadRunning_ = false; });
}

void Filter::stop()                                        // filter.cpp:419-423
{
  running_ = false;
  { std::lock_guard<std::mutex> l(bufferMutex_); }
  hasInput_.notify_all();
  if (thread_.joinable()) thread_.join();
}

void Filter::run()                                         // filter.cpp:425-443
{
  while (running_) {
    {
      std::unique_lock<std::mutex> l(bufferMutex_);
      hasInput_.wait(l, [this] { return !running_ || !inBuffer_.empty(); });
    }
    if (!running_) break;
    const auto t0 = std::chrono::steady_clock::now();
    process();
    busyNs_ += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
}

uint32_t Filter::bufferedInputs() { std::lock_guard<std::mutex> l(bufferMutex_); return (uint32_t)inBuffer_.size(); }
bool Filter::waitBufferedBelow(uint32_t n, int timeout_ms)
{
  std::unique_lock<std::mutex> l(bufferMutex_);
  return inputTaken_cv_.wait_for(l, std::chrono::milliseconds(timeout_ms), [&] { return inBuffer_.size() < n; });
}

void Filter::putInput(std::unique_ptr<Data> data)          // filter.cpp:151-222
{
  if (!data) return;
  std::lock_guard<std::mutex> l(bufferMutex_);
  ++inputTaken_;
  queueSum_ += inBuffer_.size(); ++queueSamples_;
  inBuffer_.push_back(std::move(data));
  if (maxBufferSize_ != -1 && inBuffer_.size() >= (uint32_t)maxBufferSize_) {
    if (inBuffer_[0]->type == DT_HEVCVIDEO) {
      // search for intra frames and discard everything up to it (filter.cpp:179-196)
      for (uint32_t i = 0; i < inBuffer_.size(); ++i) {
        const unsigned char *buff = inBuffer_.at(i)->data.get();
        if (!isHEVCIntra(buff)) {
          for (int j = (int)i; j != 0; --j) inBuffer_.pop_front();
          break;
        }
      }
    } else {
      inBuffer_.pop_front();                               // discard the oldest
    }
    ++inputDiscarded_;
    if (stats_) stats_->droppedPackets++;
  }
  hasInput_.notify_one();
}

std::unique_ptr<Data> Filter::getInput()                   // filter.cpp:297-306
{
  std::lock_guard<std::mutex> l(bufferMutex_);
  std::unique_ptr<Data> r;
  if (!inBuffer_.empty()) { r = std::move(inBuffer_.front()); inBuffer_.pop_front(); inputTaken_cv_.notify_all(); }
  return r;
}

Data *Filter::deepDataCopy(const Data *o)
{
  Data *c = new Data;
  c->source = o->source; c->type = o->type; c->data_size = o->data_size;
  c->creationTimestamp = o->creationTimestamp; c->presentationTimestamp = o->presentationTimestamp; c->creationUs = o->creationUs;
  c->device_data = o->device_data; c->flush_marker = o->flush_marker; c->host_view = o->host_view;
  for (int i = 0; i < 3; i++) { c->device_planes[i] = o->device_planes[i]; c->device_pitch[i] = o->device_pitch[i]; }
  if (o->data) { c->data.reset(new uint8_t[o->data_size]); memcpy(c->data.get(), o->data.get(), o->data_size); }
  if (o->vInfo) {
    c->vInfo.reset(new VideoInfo);
    c->vInfo->width = o->vInfo->width; c->vInfo->height = o->vInfo->height;
    c->vInfo->framerateNumerator = o->vInfo->framerateNumerator; c->vInfo->framerateDenominator = o->vInfo->framerateDenominator;
    c->vInfo->flippedVertically = o->vInfo->flippedVertically; c->vInfo->flippedHorizontally = o->vInfo->flippedHorizontally;
    c->vInfo->roi.width = o->vInfo->roi.width; c->vInfo->roi.height = o->vInfo->roi.height;
    if (o->vInfo->roi.data) {
      size_t n = (size_t)o->vInfo->roi.width * o->vInfo->roi.height;
      c->vInfo->roi.data.reset(new int8_t[n]); memcpy(c->vInfo->roi.data.get(), o->vInfo->roi.data.get(), n);
    }
  }
  return c;
}

void Filter::sendOutput(std::unique_ptr<Data> output)      // filter.cpp:364-417
{
  if (!output) return;
  std::lock_guard<std::mutex> l(connectionMutex_);
  if (outDataCallbacks_.empty() && outConnections_.empty()) return;
  if (!outDataCallbacks_.empty()) {
    for (size_t i = 0; i + 1 < outDataCallbacks_.size(); ++i) outDataCallbacks_[i](std::unique_ptr<Data>(deepDataCopy(output.get())));
    if (!outConnections_.empty()) outDataCallbacks_.back()(std::unique_ptr<Data>(deepDataCopy(output.get())));
    else { outDataCallbacks_.back()(std::move(output)); return; }
  }
  if (!outConnections_.empty()) {
    for (size_t i = 0; i + 1 < outConnections_.size(); ++i) outConnections_[i]->putInput(std::unique_ptr<Data>(deepDataCopy(output.get())));
    outConnections_.back()->putInput(std::move(output));   // always move the last out connection
  }
}

// ----------------------------------------------------------------------------------------------- KvazaarFilter
KvazaarFilter::KvazaarFilter(std::string id, Stats *stats, const Settings *settings)
    : Filter(std::move(id), "Kvazaar", stats, DT_YUV420VIDEO, DT_HEVCVIDEO), settings_(settings)
{
  maxBufferSize_ = 30;                                     // kvazaarfilter.cpp:28
}
KvazaarFilter::~KvazaarFilter() { stop(); close(); }

std::string KvazaarFilter::setting(const std::string &key, const std::string &def) const
{
  auto it = settings_->find(key);
  return it == settings_->end() ? def : it->second;
}

void KvazaarFilter::createInputVector(int size)            // kvazaarfilter.cpp:32-42
{
  cleanupInputVector();
  for (int i = 0; i < size; ++i) addInputPic((int)inputPics_.size());
  nextInputPic_ = 0;
}
void KvazaarFilter::cleanupInputVector()                   // kvazaarfilter.cpp:45-59
{
  if (!api_) return;
  for (auto pic : inputPics_) { pic->roi.roi_array = nullptr; api_->picture_free(pic); }
  inputPics_.clear();
  nextInputPic_ = -1;
}
void KvazaarFilter::addInputPic(int index)                 // kvazaarfilter.cpp:61-74
{
  if (!api_ || !config_) return;
  kvz_picture *pic = api_->picture_alloc(config_->width, config_->height);
  if (pic) inputPics_.insert(inputPics_.begin() + index, pic);
}
kvz_picture *KvazaarFilter::getNextPic()                   // kvazaarfilter.cpp:76-88
{
  if (encodingFrames_.size() == inputPics_.size()) addInputPic(nextInputPic_);
  kvz_picture *inputPic = inputPics_.at((size_t)nextInputPic_);
  nextInputPic_ = (nextInputPic_ + 1) % (int)inputPics_.size();
  return inputPic;
}

void KvazaarFilter::updateSettings()                       // kvazaarfilter.cpp:91-119
{
  stop();
  close();
  {
    std::lock_guard<std::mutex> l(settingsMutex_);
    init();
    encodingFrames_.clear();
  }
  start();
}

bool KvazaarFilter::init()                                 // kvazaarfilter.cpp:122-311, same order of calls
{
  if (!inputPics_.empty() || api_) return true;
  const int w = atoi(setting("video/ResolutionWidth").c_str()), h = atoi(setting("video/ResolutionHeight").c_str());
  const int fn = atoi(setting("video/FramerateNumerator").c_str()), fd = atoi(setting("video/FramerateDenominator").c_str());
  if (w == 0 || h == 0 || fn == 0 || fd == 0) { fprintf(stderr, "KvazaarFilter: invalid values in settings\n"); return false; }
  api_ = kvz_api_get(8);
  if (!api_) return false;
  config_ = api_->config_alloc();
  enc_ = nullptr;
  if (!config_) return false;
  api_->config_init(config_);
  const std::string res = std::to_string(w) + "x" + std::to_string(h), fps = std::to_string(fn) + "/" + std::to_string(fd);
  api_->config_parse(config_, "preset", setting("video/Preset", "ultrafast").c_str());
  api_->config_parse(config_, "input-res", res.c_str());
  api_->config_parse(config_, "input-fps", fps.c_str());
  std::string threads = setting("video/kvzThreads", "auto");
  if (threads == "auto") threads = std::to_string(std::thread::hardware_concurrency());
  else if (threads == "Main") threads = "0";
  api_->config_parse(config_, "threads", threads.c_str());
  api_->config_parse(config_, "owf", setting("video/OWF", "0").c_str());
  api_->config_parse(config_, "wpp", setting("video/WPP", "1").c_str());
  const bool tiles = atoi(setting("video/Tiles", "0").c_str()) != 0;
  if (tiles) api_->config_parse(config_, "tiles", setting("video/tileDimensions", "2x2").c_str());
  if (atoi(setting("video/Slices", "0").c_str()) == 1) {
    if (config_->wpp) api_->config_parse(config_, "slices", "wpp");
    else if (tiles) api_->config_parse(config_, "slices", "tiles");
  }
  api_->config_parse(config_, "qp", setting("video/QP", "32").c_str());
  api_->config_parse(config_, "period", setting("video/Intra", "64").c_str());
  api_->config_parse(config_, "vps-period", setting("video/VPS", "1").c_str());
  config_->target_bitrate = atoi(setting("video/bitrate", "0").c_str());
  if (config_->target_bitrate != 0) api_->config_parse(config_, "rc-algorithm", setting("video/rcAlgorithm", "lambda").c_str());
  api_->config_parse(config_, "intra-bits", "");
  api_->config_parse(config_, "gop", "lp-g4d3t1");
  if (atoi(setting("video/scalingList", "0").c_str()) == 0) api_->config_parse(config_, "scaling-list", "off");
  else api_->config_parse(config_, "scaling-list", "default");
  config_->lossless = atoi(setting("video/lossless", "0").c_str());
  const std::string constraint = setting("video/mvConstraint", "none");
  if (constraint == "frame" || constraint == "frametile" || constraint == "frametilemargin") api_->config_parse(config_, "mv-constraint", "");
  else api_->config_parse(config_, "mv-constraint", "none");
  if (constraint == "frame") config_->mv_constraint = KVZ_MV_CONSTRAIN_FRAME;
  else if (constraint == "tile") config_->mv_constraint = KVZ_MV_CONSTRAIN_TILE;
  else if (constraint == "frametile") config_->mv_constraint = KVZ_MV_CONSTRAIN_FRAME_AND_TILE;
  else if (constraint == "frametilemargin") config_->mv_constraint = KVZ_MV_CONSTRAIN_FRAME_AND_TILE_MARGIN;
  else config_->mv_constraint = KVZ_MV_CONSTRAIN_NONE;
  config_->set_qp_in_cu = atoi(setting("video/qpInCU", "0").c_str());
  const int vaq = atoi(setting("video/vaq", "0").c_str());
  if (vaq > 0 && vaq <= 20) api_->config_parse(config_, "vaq", setting("video/vaq").c_str());
  customParameters();
  config_->hash = KVZ_HASH_NONE;
  enc_ = api_->encoder_open(config_);
  if (!enc_) { fprintf(stderr, "KvazaarFilter: failed to open the encoder\n"); return false; }
  createInputVector(config_->owf + 1);
  { const int n = atoi(setting("uvgx/copyThreads", "4").c_str()); copy_.reset(n > 1 ? new kvzx::CopyPool(n > 16 ? 16 : n) : nullptr); }
  return !inputPics_.empty();
}

void KvazaarFilter::customParameters()                     // kvazaarfilter.cpp:351-371: INI array "parameters"
{
  const int size = atoi(setting("parameters/size", "0").c_str());
  for (int i = 1; i <= size; ++i) {
    const std::string name = setting("parameters/" + std::to_string(i) + "/Name"), value = setting("parameters/" + std::to_string(i) + "/Value");
    if (api_->config_parse(config_, name.c_str(), value.c_str()) != 1)
      fprintf(stderr, "KvazaarFilter: invalid custom parameter %s=%s\n", name.c_str(), value.c_str());
  }
}

void KvazaarFilter::close()                                // kvazaarfilter.cpp:313-329
{
  if (api_) {
    api_->encoder_close(enc_);
    api_->config_destroy(config_);
    enc_ = nullptr; config_ = nullptr;
    cleanupInputVector();
    api_ = nullptr;
  }
  pts_ = 0;
}

void KvazaarFilter::process()                              // kvazaarfilter.cpp:331-349
{
  std::unique_ptr<Data> input = getInput();
  while (input) {
    if (inputPics_.empty()) break;
    {
      std::lock_guard<std::mutex> l(settingsMutex_);
      feedInput(std::move(input));
    }
    input = getInput();
  }
}

void KvazaarFilter::feedInput(std::unique_ptr<Data> input) // kvazaarfilter.cpp:374-450
{
  kvz_picture *recon_pic = nullptr;
  kvz_frame_info frame_info;
  kvz_data_chunk *data_out = nullptr;
  uint32_t len_out = 0;
  if (input->flush_marker) { drain(); sendOutput(std::move(input)); return; }
  if (config_->width != input->vInfo->width || config_->height != input->vInfo->height ||
      config_->framerate_num != input->vInfo->framerateNumerator || config_->framerate_denom != input->vInfo->framerateDenominator) {
    fprintf(stderr, "KvazaarFilter: input resolution or framerate differs from settings\n");
    return;
  }
  if (nextInputPic_ == -1 || nextInputPic_ >= (int)inputPics_.size()) return;
  if (input->device_data) {
    // extension: the picture is already in HBM -- no kvz_picture copy, no chunk list (include/kvazzup_amd.h)
    const size_t cap = (size_t)config_->width * config_->height * 3 + (1 << 20);
    if (au_.size() < cap) au_.resize(cap);
    uint32_t n = 0;
    ++pts_;
    const void *dptr = input->device_data;
    input->device_data = nullptr;
    lastInputOnDevice_ = true;
    encodingFrames_.push_front({std::move(input), nullptr});
    // with video/OWF >= 1 the access unit that comes back belongs to the previous picture (n == 0 on the first call)
    if (!kvzx_encoder_encode_device(enc_, dptr, au_.data(), (uint32_t)au_.size(), &n, &frame_info)) { encodingFrames_.pop_front(); return; }
    if (n == 0) return;
    FrameInfo info = std::move(encodingFrames_.back());
    encodingFrames_.pop_back();
    std::unique_ptr<uint8_t[]> hevc_frame(new uint8_t[n]);
    memcpy(hevc_frame.get(), au_.data(), n);
    if (getStats()) { getStats()->encodingDelaySumMs += (uint64_t)(now_ms() - info.data->creationTimestamp); if (info.data->creationUs >= 0) getStats()->encodingDelayUs.add(now_us() - info.data->creationUs); getStats()->encodedPackets++; getStats()->encodedBytes += n; }
    sendEncodedFrame(std::move(info.data), std::move(hevc_frame), n);
    return;
  }
  lastInputOnDevice_ = false;
  kvzx::tl("feed0", pts_);
  kvz_picture *inputPic = getNextPic();
  const size_t ny = (size_t)input->vInfo->width * input->vInfo->height;
  const uint8_t *in = input->data ? input->data.get() : input->host_view;
  if (!in) return;
  if (copy_) {                                             // kvazaarfilter.cpp:410-418, the three copies shared between cores
    pieces_.clear();
    kvzx::CopyPool::add_plane(pieces_, inputPic->y, in, ny, 1, ny, ny);
    kvzx::CopyPool::add_plane(pieces_, inputPic->u, &(in[ny]), ny / 4, 1, ny / 4, ny / 4);
    kvzx::CopyPool::add_plane(pieces_, inputPic->v, &(in[ny + ny / 4]), ny / 4, 1, ny / 4, ny / 4);
    copy_->run(pieces_);
  } else {
    memcpy(inputPic->y, in, ny);                           // kvazaarfilter.cpp:410-418
    memcpy(inputPic->u, &(in[ny]), ny / 4);
    memcpy(inputPic->v, &(in[ny + ny / 4]), ny / 4);
  }
  input->host_view = nullptr;
  kvzx::tl("copied", pts_);
  inputPic->pts = pts_;
  ++pts_;
  if (config_->target_bitrate == 0) {
    inputPic->roi.width = input->vInfo->roi.width;
    inputPic->roi.height = input->vInfo->roi.height;
    inputPic->roi.roi_array = input->vInfo->roi.data.release();      // deleted after the frame's output
  }
  encodingFrames_.push_front({std::move(input), inputPic->roi.roi_array});
  api_->encoder_encode(enc_, inputPic, &data_out, &len_out, &recon_pic, nullptr, &frame_info);
  while (data_out != nullptr) {
    parseEncodedFrame(data_out, len_out, recon_pic);
    api_->encoder_encode(enc_, nullptr, &data_out, &len_out, &recon_pic, nullptr, &frame_info);
  }
}

// flush marker (harness extension): the pictures the encoder still holds (video/OWF) come out, like the reference's
// encoder_encode(NULL) loop at kvazaarfilter.cpp:440-448 run to the end
void KvazaarFilter::drain()
{
  kvz_frame_info frame_info;
  while (!encodingFrames_.empty()) {
    if (lastInputOnDevice_) {
      uint32_t n = 0;
      if (!kvzx_encoder_encode_device(enc_, nullptr, au_.data(), (uint32_t)au_.size(), &n, &frame_info) || n == 0) break;
      FrameInfo info = std::move(encodingFrames_.back());
      encodingFrames_.pop_back();
      std::unique_ptr<uint8_t[]> hevc_frame(new uint8_t[n]);
      memcpy(hevc_frame.get(), au_.data(), n);
      if (getStats()) { getStats()->encodingDelaySumMs += (uint64_t)(now_ms() - info.data->creationTimestamp); if (info.data->creationUs >= 0) getStats()->encodingDelayUs.add(now_us() - info.data->creationUs); getStats()->encodedPackets++; getStats()->encodedBytes += n; }
      sendEncodedFrame(std::move(info.data), std::move(hevc_frame), n);
    } else {
      kvz_picture *recon_pic = nullptr; kvz_data_chunk *data_out = nullptr; uint32_t len_out = 0;
      api_->encoder_encode(enc_, nullptr, &data_out, &len_out, &recon_pic, nullptr, &frame_info);
      if (!data_out) {
        if (config_->null_input_poll && kvzx_encoder_pending(enc_) > 0) { std::this_thread::sleep_for(std::chrono::microseconds(50)); continue; }   // (null-input=poll: not finished yet)
        break;
      }
      parseEncodedFrame(data_out, len_out, recon_pic);
    }
  }
}

void KvazaarFilter::parseEncodedFrame(kvz_data_chunk *data_out, uint32_t len_out, kvz_picture *recon_pic)   // kvazaarfilter.cpp:453-484
{
  FrameInfo info = std::move(encodingFrames_.back());
  encodingFrames_.pop_back();
  if (info.roi_array) { delete[] info.roi_array; info.roi_array = nullptr; }
  std::unique_ptr<uint8_t[]> hevc_frame(new uint8_t[len_out]);
  uint8_t *writer = hevc_frame.get();
  uint32_t dataWritten = 0;
  for (kvz_data_chunk *chunk = data_out; chunk != nullptr; chunk = chunk->next) {
    memcpy(writer, chunk->data, chunk->len);
    writer += chunk->len; dataWritten += chunk->len;
  }
  api_->chunk_free(data_out);
  api_->picture_free(recon_pic);
  if (getStats()) { getStats()->encodingDelaySumMs += (uint64_t)(now_ms() - info.data->creationTimestamp); if (info.data->creationUs >= 0) getStats()->encodingDelayUs.add(now_us() - info.data->creationUs); getStats()->encodedPackets++; getStats()->encodedBytes += len_out; }
  sendEncodedFrame(std::move(info.data), std::move(hevc_frame), dataWritten);
}

void KvazaarFilter::sendEncodedFrame(std::unique_ptr<Data> input, std::unique_ptr<uint8_t[]> hevc_frame, uint32_t dataWritten)   // :487-495
{
  input->type = DT_HEVCVIDEO;
  input->data_size = dataWritten;
  input->data = std::move(hevc_frame);
  sendOutput(std::move(input));
}

// ----------------------------------------------------------------------------------------------- OpenHEVCFilter
enum OHThreadType { OH_THREAD_FRAME = 1, OH_THREAD_SLICE = 2, OH_THREAD_FRAMESLICE = 3 };

OpenHEVCFilter::OpenHEVCFilter(uint32_t sessionID, Stats *stats, const Settings *settings)
    : Filter(std::to_string(sessionID), "OpenHEVC", stats, DT_HEVCVIDEO, DT_YUV420VIDEO), settings_(settings), sessionID_(sessionID) {}
OpenHEVCFilter::~OpenHEVCFilter()
{
  stop();
  finishOutput();
  if (handle_) uninit();
}

// the output stage's thread copies and sends what is still queued, then ends (the frames' memory lives until uninit).  Whoever owns what the
// output callbacks touch calls this before that goes away (uvgx_pipeline_destroy).
void OpenHEVCFilter::finishOutput()
{
  if (!outThread_.joinable()) return;
  { std::lock_guard<std::mutex> l(outM_); outQuit_ = true; }
  outCv_.notify_all();
  outThread_.join();
  outQuit_ = false;
}

bool OpenHEVCFilter::init()                                // openhevcfilter.cpp:28-74
{
  auto get = [&](const char *k, const char *d) { auto it = settings_->find(k); return it == settings_->end() ? std::string(d) : it->second; };
  threads_ = atoi(get("video/OPENHEVC_threads", "1").c_str());
  parallelizationMode_ = get("video/OH_parallelization", "Slice");
  if (parallelizationMode_ == "Slice") handle_ = libOpenHevcInit(threads_, OH_THREAD_SLICE);
  else if (parallelizationMode_ == "Frame") handle_ = libOpenHevcInit(threads_, OH_THREAD_FRAME);
  else handle_ = libOpenHevcInit(threads_, OH_THREAD_FRAMESLICE);
  const std::string dev = get("uvgx/gpu", "");
  if (!dev.empty()) kvzx_decoder_set_device(handle_, atoi(dev.c_str()));
  if (libOpenHevcStartDecoder(handle_) == -1) { fprintf(stderr, "OpenHEVCFilter: failed to start decoder\n"); return false; }
  libOpenHevcSetTemporalLayer_id(handle_, 0);
  libOpenHevcSetActiveDecoders(handle_, 0);
  libOpenHevcSetViewLayers(handle_, 0);
  download_ = atoi(get("uvgx/decoderDownload", "1").c_str()) != 0;
  { const int n = atoi(get("uvgx/copyThreads", "4").c_str()); copy_.reset(n > 1 && download_ ? new kvzx::CopyPool(n > 16 ? 16 : n) : nullptr); }
  asyncOut_ = atoi(get("uvgx/asyncOutput", "1").c_str()) != 0;
  if (!download_) kvzx_decoder_set_download(handle_, 0);   // extension: leave decoded pictures in HBM
  decodingFrames_.clear();
  maxBufferSize_ = -1;                                     // no buffer limit (openhevcfilter.cpp:68)
  vpsReceived_ = spsReceived_ = ppsReceived_ = false;
  return true;
}

void OpenHEVCFilter::uninit()                              // openhevcfilter.cpp:77-83
{
  libOpenHevcFlush(handle_);
  libOpenHevcClose(handle_);
  handle_ = nullptr;
}

void OpenHEVCFilter::updateSettings()                      // openhevcfilter.cpp:86-100
{
  auto get = [&](const char *k, const char *d) { auto it = settings_->find(k); return it == settings_->end() ? std::string(d) : it->second; };
  if (atoi(get("video/OPENHEVC_threads", "1").c_str()) != threads_ || get("video/OH_parallelization", "Slice") != parallelizationMode_) {
    std::lock_guard<std::mutex> l(settingsMutex_);
    finishOutput();                                        // (pictures still being copied out live in the decoder that is about to go)
    uninit();
    init();
  }
}

void OpenHEVCFilter::process()                             // openhevcfilter.cpp:103-189
{
  std::unique_ptr<Data> input = getInput();
  while (input) {
    if (getStats()) { getStats()->receivedPackets++; getStats()->receivedBytes += input->data_size; }
    if (!input->data || input->data_size < 6) { input = getInput(); continue; }      // (not a NAL unit -- nothing behind the start code; the reference reads buff[4] whatever the size, openhevcfilter.cpp:114)
    {
      std::lock_guard<std::mutex> l(settingsMutex_);
      const unsigned char *buff = input->data.get();
      uint8_t nalType = (buff[4] >> 1);
      if (!vpsReceived_ && nalType == VPS_NUT) vpsReceived_ = true;
      if (!spsReceived_ && nalType == SPS_NUT) spsReceived_ = true;
      if (!ppsReceived_ && nalType == PPS_NUT) ppsReceived_ = true;
      bool vcl = nalType <= 31;
      if ((vpsReceived_ && spsReceived_ && ppsReceived_) || !vcl) {
        discardedFrames_ = 0;
        kvzx::tl("dec0", (long)input->presentationTimestamp);
        if (asyncOut_) {
          // A picture's memory stays valid for a number of further decode CALLS (Decoder::kOutHold / kOutRing), whether or not they return pictures -- and at the
          // end of a stream many do not (end-of-sequence units while the frame threads finish).  So a call waits while the oldest copy still pending is of a
          // picture handed out five calls ago: the output stage may lag by pictures, never by calls.  (Found as 27 wrong bytes at the start of one picture of a
          // reordered stream under a loaded host: the copy read a buffer the allocator had taken back.)
          std::unique_lock<std::mutex> l(outM_);
          outSpace_.wait(l, [this] { return outQ_.empty() || decodeCalls_ - outQ_.front().call_no < 5; });
        }
        decodeCalls_++;
        int gotPicture = libOpenHevcDecode(handle_, input->data.get(), (int)input->data_size, input->presentationTimestamp);
        kvzx::tl("dec1", (long)input->presentationTimestamp);
        if (vcl) decodingFrames_.push_front(std::move(input));
        if (gotPicture <= -1) fprintf(stderr, "OpenHEVCFilter: error while decoding (%d)\n", kvzx_decoder_last_error(handle_));
        else if (gotPicture > 0) sendDecodedOutput(gotPicture);
      } else {
        ++discardedFrames_;
      }
    }
    input = getInput();
  }
}

void OpenHEVCFilter::sendDecodedOutput(int &gotPicture)    // openhevcfilter.cpp:192-239
{
  OpenHevc_Frame openHevcFrame;
  if ((gotPicture = libOpenHevcGetOutput(handle_, gotPicture, &openHevcFrame)) > 0) {
    std::unique_ptr<Data> decodedFrame = std::move(decodingFrames_.back());
    decodingFrames_.pop_back();
    libOpenHevcGetPictureInfo(handle_, &openHevcFrame.frameInfo);
    decodedFrame->vInfo->width = (int16_t)openHevcFrame.frameInfo.nWidth;
    decodedFrame->vInfo->height = (int16_t)openHevcFrame.frameInfo.nHeight;
    const int W = decodedFrame->vInfo->width, H = decodedFrame->vInfo->height;
    decodedFrame->type = DT_YUV420VIDEO;
    decodedFrame->vInfo->framerateNumerator = openHevcFrame.frameInfo.frameRate.num;
    decodedFrame->vInfo->framerateDenominator = openHevcFrame.frameInfo.frameRate.den;
    if (!download_) {
      // extension: the picture stays in HBM; hand its device pointer on instead of copying rows
      const void *planes[3]; int pitches[3];
      kvzx_decoder_output_device(handle_, planes, pitches);
      decodedFrame->device_data = planes[0];
      for (int i = 0; i < 3; i++) { decodedFrame->device_planes[i] = planes[i]; decodedFrame->device_pitch[i] = pitches[i]; }
      decodedFrame->data.reset();
      decodedFrame->data_size = 0;
      if (getStats() && decodedFrame->creationUs >= 0) getStats()->totalDelayUs.add(now_us() - decodedFrame->creationUs);
      sendOutput(std::move(decodedFrame));
      return;
    }
    OutJob job;
    job.y = (const uint8_t *)openHevcFrame.pvY; job.u = (const uint8_t *)openHevcFrame.pvU; job.v = (const uint8_t *)openHevcFrame.pvV;
    job.s_stride = (uint32_t)openHevcFrame.frameInfo.nYPitch; job.qs_stride = (uint32_t)openHevcFrame.frameInfo.nUPitch / 2; job.W = W; job.H = H;
    job.frame = std::move(decodedFrame); job.call_no = decodeCalls_;
    if (!asyncOut_) { copyOut(job); return; }
    if (!outThread_.joinable()) outThread_ = std::thread([this] { kvzx::name_this_thread("kvzx-dec-out"); outputStage(); });
    std::unique_lock<std::mutex> l(outM_);
    outSpace_.wait(l, [this] { return outQ_.size() < 3; });
    outQ_.push_back(std::move(job));
    outCv_.notify_one();
  }
}

void OpenHEVCFilter::outputStage()
{
  for (;;) {
    OutJob job;
    {
      std::unique_lock<std::mutex> l(outM_);
      outCv_.wait(l, [this] { return outQuit_ || !outQ_.empty(); });
      if (outQ_.empty()) return;
      job = std::move(outQ_.front());                      // (stays counted until the copy is done: the queue bound is what keeps the frame's memory valid)
    }
    copyOut(job);
    { std::lock_guard<std::mutex> l(outM_); outQ_.pop_front(); }
    outSpace_.notify_one();
  }
}

void OpenHEVCFilter::copyOut(OutJob &job)                  // the copy of openhevcfilter.cpp:206-235
{
  std::unique_ptr<Data> decodedFrame = std::move(job.frame);
  const int W = job.W, H = job.H;
  {
    uint32_t finalDataSize = (uint32_t)(W * H + W * H / 2);
    kvzx::tl("out0", (long)decodedFrame->presentationTimestamp);
    std::unique_ptr<uint8_t[]> yuv_frame(new uint8_t[finalDataSize]);
    kvzx::tl("alloc", (long)decodedFrame->presentationTimestamp);
    uint8_t *pY = yuv_frame.get(), *pU = yuv_frame.get() + W * H, *pV = yuv_frame.get() + W * H + W * H / 4;
    uint32_t s_stride = job.s_stride, qs_stride = job.qs_stride;
    uint32_t d_stride = (uint32_t)W / 2, dd_stride = (uint32_t)W;
    if (copy_) {                                           // the same rows (openhevcfilter.cpp:212-229), shared between cores
      pieces_.clear();
      kvzx::CopyPool::add_plane(pieces_, pY, job.y, dd_stride, (size_t)H, dd_stride, s_stride);
      kvzx::CopyPool::add_plane(pieces_, pU, job.u, d_stride, (size_t)((H + 1) / 2), d_stride, 2 * qs_stride);
      kvzx::CopyPool::add_plane(pieces_, pV, job.v, d_stride, (size_t)((H + 1) / 2), d_stride, 2 * qs_stride);
      copy_->run(pieces_);
    } else
    for (int i = 0; i < H; i++) {
      memcpy(pY, job.y + (size_t)i * s_stride, dd_stride);
      pY += dd_stride;
      if (!(i % 2)) {
        memcpy(pU, job.u + (size_t)i * qs_stride, d_stride); pU += d_stride;
        memcpy(pV, job.v + (size_t)i * qs_stride, d_stride); pV += d_stride;
      }
    }
    decodedFrame->data_size = finalDataSize;
    decodedFrame->data = std::move(yuv_frame);
    kvzx::tl("out1", (long)decodedFrame->presentationTimestamp);
    if (getStats() && decodedFrame->creationUs >= 0) getStats()->totalDelayUs.add(now_us() - decodedFrame->creationUs);
    sendOutput(std::move(decodedFrame));
    kvzx::tl("out2", 0);
  }
}

// ----------------------------------------------------------------------------------------------- WireAdapter (row f2)
void WireAdapter::process()
{
  std::unique_ptr<Data> input = getInput();
  while (input) {
    if (input->flush_marker) {
      // end-of-sequence NAL units (type 36): each makes the decoder hand out one picture its frame threads still hold
      for (int k = 0; k < 34; k++) {                     // (more than the decoder's largest ring: 32 frame threads)
        std::unique_ptr<Data> nal(new Data);
        nal->source = DS_REMOTE; nal->type = DT_HEVCVIDEO; nal->data_size = 6;
        nal->data.reset(new uint8_t[6]{0, 0, 0, 1, 36 << 1, 1});
        nal->vInfo.reset(new VideoInfo);
        sendOutput(std::move(nal));
      }
      input = getInput();
      continue;
    }
    const uint8_t *p = input->data.get(); const uint32_t n = input->data_size;
    std::vector<uint32_t> starts;
    for (uint32_t i = 0; i + 3 < n;) {                     // 00 00 00 01: from zero byte to zero byte (memchr), not byte by byte
      const uint8_t *z = (const uint8_t *)memchr(p + i, 0, n - 3 - i);
      if (!z) break;
      i = (uint32_t)(z - p);
      if (p[i + 1] == 0 && p[i + 2] == 0 && p[i + 3] == 1) { starts.push_back(i); i += 4; } else i++;
    }
    starts.push_back(n);
    if (lossEvery_ > 0) {
      bool precious = false;                                 // parameter sets, IRAP pictures
      for (size_t k = 0; k + 1 < starts.size(); k++) { const int t = (p[starts[k] + 4] >> 1) & 0x3f; precious |= (t >= 16 && t <= 23) || (t >= 32 && t <= 34); }
      if (!precious && ++seen_ % (uint64_t)lossEvery_ == 0) { lost_.fetch_add(1); input = getInput(); continue; }
    }
    for (size_t k = 0; k + 1 < starts.size(); k++) {
      std::unique_ptr<Data> nal(new Data);
      nal->source = DS_REMOTE; nal->type = DT_HEVCVIDEO;
      nal->data_size = starts[k + 1] - starts[k];
      nal->data.reset(new uint8_t[nal->data_size]);
      memcpy(nal->data.get(), p + starts[k], nal->data_size);
      nal->creationTimestamp = input->creationTimestamp; nal->creationUs = input->creationUs;
      nal->presentationTimestamp = input->presentationTimestamp;
      nal->vInfo.reset(new VideoInfo);                     // resolution unknown until decoded (filter.cpp initializeData)
      sendOutput(std::move(nal));
    }
    input = getInput();
  }
}

// ----------------------------------------------------------------------------------------------- YUVtoRGB32
YUVtoRGB32::~YUVtoRGB32() { stop(); for (void *p : ring_) if (p) hipFree(p); }

void YUVtoRGB32::process()                                  // yuvtorgb32.cpp:29-64
{
  std::unique_ptr<Data> input = getInput();
  while (input) {
    const int W = input->vInfo->width, H = input->vInfo->height;
    const uint32_t finalDataSize = (uint32_t)(W * H * 4);
    if (input->device_planes[0]) {
      // the decoder left the picture in HBM: convert it there
      if (ring_bytes_ != finalDataSize) {
        for (void *&p : ring_) { if (p) hipFree(p); p = nullptr; }
        ring_bytes_ = finalDataSize;
      }
      void *&dst = ring_[next_]; next_ = (next_ + 1) % kRing;
      if (!dst && hipMalloc(&dst, ring_bytes_) != hipSuccess) { dst = nullptr; input = getInput(); continue; }
      if (!kvzx_yuv420_to_rgb32_device(input->device_planes[0], input->device_planes[1], input->device_planes[2], input->device_pitch[0],
                                       input->device_pitch[1], dst, W, H, 0, nullptr) || hipStreamSynchronize(nullptr) != hipSuccess) {
        fprintf(stderr, "YUVtoRGB32: conversion failed\n");
        input = getInput();
        continue;
      }
      input->device_data = dst;
      for (int i = 0; i < 3; i++) { input->device_planes[i] = nullptr; input->device_pitch[i] = 0; }
      input->data.reset(); input->data_size = 0;
    } else {
      std::unique_ptr<uint8_t[]> rgb32_frame(new uint8_t[finalDataSize]);
      if (!kvzx_yuv420_to_rgb32(input->data.get(), rgb32_frame.get(), W, H, 0)) { fprintf(stderr, "YUVtoRGB32: conversion failed\n"); input = getInput(); continue; }
      input->data = std::move(rgb32_frame);
      input->data_size = finalDataSize;
    }
    input->type = DT_RGB32VIDEO;
    sendOutput(std::move(input));
    input = getInput();
  }
}

}  // namespace uvgx

using namespace uvgx;

// ----------------------------------------------------------------------------------------------- C shim for tests / bench
struct UvgxPipeline {
  Settings settings;
  Stats stats;
  std::unique_ptr<KvazaarFilter> enc;
  std::unique_ptr<WireAdapter> wire;
  std::unique_ptr<OpenHEVCFilter> dec;
  std::unique_ptr<YUVtoRGB32> rgb;                         // settings uvgx/rgb32Output=1: the display-side conversion
  std::mutex m; std::condition_variable cv;
  std::deque<std::unique_ptr<Data>> encoded, decoded;
  uint64_t n_encoded = 0, n_decoded = 0, decoded_bytes = 0;
  std::vector<uint32_t> lat_enc, lat_total;              // per picture, us: pushed -> access unit out of the encoder filter; -> decoded picture out of the last filter
  bool keep = true, loopback = true;
};

extern "C" {

// settings_text: "key=value" lines with the key names of src/settingskeys.h (plus uvgx/* extensions).
KVZ_PUBLIC void *uvgx_pipeline_create(const char *settings_text, int loopback_decode, int keep_outputs)
{
  UvgxPipeline *p = new UvgxPipeline();
  p->keep = keep_outputs != 0; p->loopback = loopback_decode != 0;
  std::string s(settings_text ? settings_text : "");
  size_t pos = 0;
  while (pos < s.size()) {
    size_t e = s.find('\n', pos); if (e == std::string::npos) e = s.size();
    std::string line = s.substr(pos, e - pos); pos = e + 1;
    size_t eq = line.find('=');
    if (eq != std::string::npos) p->settings[line.substr(0, eq)] = line.substr(eq + 1);
  }
  p->enc.reset(new KvazaarFilter("uvgx", &p->stats, &p->settings));
  if (!p->enc->init()) { delete p; return nullptr; }
  p->enc->addDataOutCallback([p](std::unique_ptr<Data> d) {
    if (d->flush_marker) return;
    const int64_t lat = d->creationUs >= 0 ? now_us() - d->creationUs : -1;
    std::lock_guard<std::mutex> l(p->m);
    p->n_encoded++;
    if (lat >= 0 && p->lat_enc.size() < (1u << 20)) p->lat_enc.push_back((uint32_t)(lat > 0xffffffffll ? 0xffffffffll : lat));
    if (p->keep) p->encoded.push_back(std::move(d));
    p->cv.notify_all();
  });
  if (p->loopback) {
    p->wire.reset(new WireAdapter("uvgx", &p->stats));
    { auto it = p->settings.find("uvgx/wireLossEvery"); if (it != p->settings.end()) p->wire->setLossEvery(atoi(it->second.c_str())); }
    p->dec.reset(new OpenHEVCFilter(1, &p->stats, &p->settings));
    if (!p->dec->init()) { delete p; return nullptr; }
    p->enc->addOutConnection(p->wire.get());
    p->wire->addOutConnection(p->dec.get());
    Filter *last = p->dec.get();
    auto it = p->settings.find("uvgx/rgb32Output");
    if (it != p->settings.end() && atoi(it->second.c_str()) != 0) {
      p->rgb.reset(new YUVtoRGB32("uvgx", &p->stats));
      p->dec->addOutConnection(p->rgb.get());
      last = p->rgb.get();
    }
    last->addDataOutCallback([p](std::unique_ptr<Data> d) {
      const int64_t lat = d->creationUs >= 0 ? now_us() - d->creationUs : -1;
      std::lock_guard<std::mutex> l(p->m);
      p->n_decoded++; p->decoded_bytes += d->data_size;
      if (lat >= 0 && p->lat_total.size() < (1u << 20)) p->lat_total.push_back((uint32_t)(lat > 0xffffffffll ? 0xffffffffll : lat));
      if (p->keep) p->decoded.push_back(std::move(d));
      p->cv.notify_all();
    });
    p->wire->start();
    p->dec->start();
    if (p->rgb) p->rgb->start();
  }
  p->enc->start();
  return p;
}

static int push(UvgxPipeline *p, const uint8_t *host, const void *dev, int w, int h, int fn, int fd, int64_t pts, bool borrow = false)
{
  std::unique_ptr<Data> d(new Data);
  d->source = DS_LOCAL; d->type = DT_YUV420VIDEO;
  d->creationTimestamp = now_ms(); d->creationUs = now_us(); d->presentationTimestamp = pts;
  d->vInfo.reset(new VideoInfo);
  d->vInfo->width = (int16_t)w; d->vInfo->height = (int16_t)h; d->vInfo->framerateNumerator = fn; d->vInfo->framerateDenominator = fd;
  if (host && borrow) { d->data_size = (uint32_t)(w * h * 3 / 2); d->host_view = host; }
  else if (host) { d->data_size = (uint32_t)(w * h * 3 / 2); d->data.reset(new uint8_t[d->data_size]); memcpy(d->data.get(), host, d->data_size); }
  d->device_data = dev;
  p->enc->putInput(std::move(d));
  return 1;
}
KVZ_PUBLIC int uvgx_pipeline_push_host(void *pp, const uint8_t *i420, int w, int h, int fn, int fd, int64_t pts) { return pp && i420 ? push((UvgxPipeline *)pp, i420, nullptr, w, h, fn, fd, pts) : 0; }
KVZ_PUBLIC int uvgx_pipeline_push_device(void *pp, const void *d_i420, int w, int h, int fn, int fd, int64_t pts) { return pp && d_i420 ? push((UvgxPipeline *)pp, nullptr, d_i420, w, h, fn, fd, pts) : 0; }

// measurement aid: an access unit straight into the receiving side (WireAdapter -> OpenHEVCFilter), paced like the sources above; size 0 = flush marker
KVZ_PUBLIC int uvgx_pipeline_push_encoded(void *pp, const uint8_t *au, uint32_t size, int64_t pts, uint32_t max_backlog, int timeout_ms)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  if (!p || !p->wire || !p->dec) return 0;
  std::unique_ptr<Data> d(new Data);
  d->source = DS_REMOTE; d->type = DT_HEVCVIDEO; d->creationTimestamp = now_ms(); d->presentationTimestamp = pts;
  if (!size) d->flush_marker = true;
  else {
    if (!p->dec->waitBufferedBelow(max_backlog, timeout_ms)) return 0;
    d->data_size = size; d->data.reset(new uint8_t[size]); memcpy(d->data.get(), au, size);
    d->vInfo.reset(new VideoInfo);
  }
  p->wire->putInput(std::move(d));
  return 1;
}

// wait until `n` pictures have left the last filter (decoder when looped back, else encoder); 1 = reached
KVZ_PUBLIC int uvgx_pipeline_wait(void *pp, uint64_t n, int timeout_ms)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  std::unique_lock<std::mutex> l(p->m);
  return p->cv.wait_for(l, std::chrono::milliseconds(timeout_ms), [&] { return (p->loopback ? p->n_decoded : p->n_encoded) >= n; }) ? 1 : 0;
}
// everything pushed so far comes out without further input (then uvgx_pipeline_wait for the count)
KVZ_PUBLIC int uvgx_pipeline_flush(void *pp)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  if (!p) return 0;
  std::unique_ptr<Data> d(new Data);
  d->flush_marker = true;
  p->enc->putInput(std::move(d));
  return 1;
}
KVZ_PUBLIC uint32_t uvgx_pipeline_encoder_backlog(void *pp) { return ((UvgxPipeline *)pp)->enc->bufferedInputs(); }
// a source that paces itself: sleeps until the encoder filter buffers fewer than `max_backlog` pictures (a uvgComm filter drops
// inputs at 10 buffered, filter.cpp:151-222), then pushes; 0 = timed out or push failed
KVZ_PUBLIC int uvgx_pipeline_push_device_paced(void *pp, const void *d_i420, int w, int h, int fn, int fd, int64_t pts, uint32_t max_backlog, int timeout_ms)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  if (!p || !d_i420) return 0;
  if (!p->enc->waitBufferedBelow(max_backlog, timeout_ms)) return 0;
  return push(p, nullptr, d_i420, w, h, fn, fd, pts);
}

// the same for a host picture (what uvgComm's graph hands the encoder filter).  borrow != 0: the Data refers to the caller's buffer, which must
// stay unchanged until the picture has been encoded (the harness's clip); 0: the Data owns a copy, like a camera filter's output
KVZ_PUBLIC int uvgx_pipeline_push_host_paced(void *pp, const uint8_t *i420, int w, int h, int fn, int fd, int64_t pts, uint32_t max_backlog, int timeout_ms, int borrow)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  if (!p || !i420) return 0;
  if (!p->enc->waitBufferedBelow(max_backlog, timeout_ms)) return 0;
  return push(p, i420, nullptr, w, h, fn, fd, pts, borrow != 0);
}

static int pop(UvgxPipeline *p, std::deque<std::unique_ptr<Data>> &q, uint8_t *buf, uint32_t cap, uint32_t *size, int *w, int *h, int64_t *pts)
{
  std::lock_guard<std::mutex> l(p->m);
  if (q.empty()) return 0;
  Data *d = q.front().get();
  if (size) *size = d->data_size;
  if (w) *w = d->vInfo ? d->vInfo->width : 0;
  if (h) *h = d->vInfo ? d->vInfo->height : 0;
  if (pts) *pts = d->presentationTimestamp;
  if (d->data_size > cap) return -1;
  if (d->data) memcpy(buf, d->data.get(), d->data_size);
  q.pop_front();
  return 1;
}
KVZ_PUBLIC int uvgx_pipeline_pop_encoded(void *pp, uint8_t *buf, uint32_t cap, uint32_t *size, int64_t *pts) { UvgxPipeline *p = (UvgxPipeline *)pp; return pop(p, p->encoded, buf, cap, size, nullptr, nullptr, pts); }
KVZ_PUBLIC int uvgx_pipeline_pop_decoded(void *pp, uint8_t *buf, uint32_t cap, uint32_t *size, int *w, int *h, int64_t *pts) { UvgxPipeline *p = (UvgxPipeline *)pp; return pop(p, p->decoded, buf, cap, size, w, h, pts); }

// out[0..7]: encoded pictures, encoded bytes, NAL units received by the decoder, their bytes, dropped inputs,
// decoded pictures, sum of encoding delays (ms), decoder-side discarded inputs
KVZ_PUBLIC void uvgx_pipeline_stats(void *pp, uint64_t *out)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  std::lock_guard<std::mutex> l(p->m);
  out[0] = p->stats.encodedPackets; out[1] = p->stats.encodedBytes; out[2] = p->stats.receivedPackets; out[3] = p->stats.receivedBytes;
  out[4] = p->stats.droppedPackets; out[5] = p->n_decoded; out[6] = p->stats.encodingDelaySumMs; out[7] = p->enc->inputDiscarded();
}
// Delay samples of the pictures that have come out since the last reset, in the order they came out, microseconds.  which 0: encoding delay (picture
// pushed -> access unit out of KvazaarFilter', what kvazaarfilter.cpp:478-479 reports), 1: total delay (-> decoded picture out of the last filter,
// displayfilter.cpp:113-115).  Returns the number of samples held; up to `cap` are copied.
KVZ_PUBLIC uint32_t uvgx_pipeline_latency_us(void *pp, int which, uint32_t *out, uint32_t cap, int reset)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  if (!p) return 0;
  std::lock_guard<std::mutex> l(p->m);
  std::vector<uint32_t> &v = which ? p->lat_total : p->lat_enc;
  const uint32_t n = (uint32_t)v.size();
  if (out) for (uint32_t i = 0; i < n && i < cap; i++) out[i] = v[i];
  if (reset) v.clear();
  return n;
}
// the filters' own statistics (uvgx::Stats histograms): out[0..3] = count, mean, p50, p99 of the encoding delay, out[4..7] of the total delay (us)
KVZ_PUBLIC void uvgx_pipeline_delay_stats(void *pp, double *out8)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  const DelayHist *h[2] = {&p->stats.encodingDelayUs, &p->stats.totalDelayUs};
  for (int k = 0; k < 2; k++) {
    const uint64_t c = h[k]->count.load();
    out8[4 * k] = (double)c; out8[4 * k + 1] = c ? (double)h[k]->sumUs.load() / (double)c : 0.0;
    out8[4 * k + 2] = h[k]->percentile(0.5); out8[4 * k + 3] = h[k]->percentile(0.99);
  }
}
KVZ_PUBLIC void uvgx_pipeline_busy_ms(void *pp, double *out3)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  out3[0] = p->enc->busyNs() * 1e-6; out3[1] = p->wire ? p->wire->busyNs() * 1e-6 : 0.0; out3[2] = p->dec ? p->dec->busyNs() * 1e-6 : 0.0;
}
KVZ_PUBLIC void uvgx_pipeline_avg_queue(void *pp, double *out3)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  out3[0] = p->enc->avgQueue(); out3[1] = p->wire ? p->wire->avgQueue() : 0.0; out3[2] = p->dec ? p->dec->avgQueue() : 0.0;
}
KVZ_PUBLIC void *uvgx_pipeline_encoder(void *pp) { return ((UvgxPipeline *)pp)->enc->encoder(); }
KVZ_PUBLIC void *uvgx_pipeline_decoder(void *pp) { UvgxPipeline *p = (UvgxPipeline *)pp; return p->dec ? p->dec->handle() : nullptr; }
KVZ_PUBLIC void uvgx_pipeline_destroy(void *pp)
{
  UvgxPipeline *p = (UvgxPipeline *)pp;
  if (!p) return;
  p->enc->stop();
  if (p->wire) p->wire->stop();
  if (p->dec) { p->dec->stop(); p->dec->finishOutput(); }
  if (p->rgb) p->rgb->stop();
  delete p;
}

}  // extern "C"
