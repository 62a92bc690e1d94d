// capi.cpp -- the C ABI of include/mijpeg.h: decoder object (host entropy decoding + streaming upload +
// GPU reconstruction + rectangle service) and the stateless batch launch.  Compiled with hipcc (host side
// only uses the HIP runtime API).  There is NO CPU fallback for the reconstruction: without a device the
// reconstruct calls fail with MIJPEG_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>

#include "../../include/mijpeg.h"
#include "host_decoder.hpp"
#include "huffman_dev.hpp"
#include "encoder.hpp"
#include "forward.hpp"
#include "hencode.hpp"
#include "kernels.hpp"
#include "request_model.hpp"

using namespace mij;

struct mijpeg_decoder {
  int device = -1;
  HostDecoder host;
  const uint8_t *data = nullptr;
  size_t size = 0;
  bool parsed = false, decoded = false, uploaded = false;
  bool parse_fresh = false; // host holds a full parse of (data, size) that nothing has touched since: mijpeg_decode_coefficients_device
                            // found the stream not to qualify, the host decode that follows need not parse again
  // coefficient store: pinned when a device is attached
  int16_t *coef_host = nullptr;
  size_t coef_host_cap = 0; // int16 units
  int16_t *coef_dev = nullptr;
  size_t coef_dev_cap = 0;
  // reconstruction cache for the rectangle service
  uint8_t *img_dev = nullptr;
  size_t img_dev_cap = 0;
  uint8_t *img_host = nullptr; // pinned
  size_t img_host_cap = 0;
  bool img_valid = false;      // img_dev holds the reconstructed frame for img_flags
  bool img_host_valid = false; // ... and img_host its copy (being filled band by band, see band_events)
  // the device-to-host copy of the reconstructed frame travels in bands of lines, one event each: a rectangle request
  // waits for the bands it touches only, so the first stripes of a frame are served while the rest is still on its way
  std::vector<hipEvent_t> band_events;
  int band_lines = 0, bands = 0, bands_waited = 0;
  uint32_t img_flags = 0;
  int img_view = -1;           // component of a non-upsampled reconstruction, -1: the whole picture
  // mijpeg_display_rect: the reference's state between DisplayRectangle calls (request_model.hpp) and the buffers of the
  // requests that do not show the plain picture
  RequestModel model, rmodel; // (rmodel: the residual image of a JPEG XT frame)
  bool model_valid = false;
  uint8_t *req_dev = nullptr, *req_host = nullptr; // frame-sized interleaved image (device; pinned host)
  size_t req_dev_cap = 0, req_host_cap = 0;
  int32_t *rowmap_dev = nullptr;
  size_t rowmap_cap = 0;
  int32_t *ws_dev = nullptr;
  size_t ws_cap = 0; // bytes
  // on-device entropy decoding: stream bytes, interval offsets, tables, status word
  uint8_t *ent_dev = nullptr;
  size_t ent_cap = 0;
  uint8_t *ent_host = nullptr; // pinned staging for offsets + tables + status
  size_t ent_host_cap = 0;
  bool host_planes_stale = false; // coefficients live on the device only
  double phase_prepare = 0, phase_device = 0; // last device entropy decode: host tables / upload + kernel
  mijpeg_decoder *xt_helper = nullptr; // JPEG XT: second context that entropy-decodes the residual codestream concurrently
  // JPEG XT alpha channel: an image of its own (ALFA box), decoded by a decoder object of its own that this one owns
  // (mijpeg_alpha_channel); its codestream is copied here because every parse of the file rebuilds the boxes
  mijpeg_decoder *alpha = nullptr;
  bool alpha_ready = false;
  int alpha_refusal = 0;          // the alpha image reads, its transformer would not build (or this path declines it): the code
  std::string alpha_refusal_msg;
  std::vector<uint8_t> alpha_data;
  uint8_t *enc_dev = nullptr; // encoder direction: pixels + coefficients of one picture
  size_t enc_cap = 0;
  uint8_t *henc_dev[2] = {nullptr, nullptr}, *henc_out_dev[2] = {nullptr, nullptr}; // device entropy coder, two jobs: arrays; streams
  size_t henc_cap[2] = {0, 0}, henc_out_cap[2] = {0, 0};
  uint64_t *henc_host = nullptr; // pinned: byte counts read back from the device, code tables on their way up
  uint8_t *walk_dev = nullptr, *walk_host = nullptr; // state of the device walk over streams without restart markers
  size_t walk_cap = 0, walk_host_cap = 0;
  int walk_rounds = 0;
  uint32_t *walk_status_dev = nullptr;
  hipStream_t copy_stream = nullptr;          // uploads of a batch's streams, ahead of the kernels that decode them
  hipEvent_t ent_free = nullptr;              // behind the last kernel / copy that reads ent_dev
  bool ent_free_valid = false;
  std::vector<hipEvent_t> copy_events;
  uint8_t *stage_host = nullptr;  // pinned gathering area for the streams of a batch
  std::vector<uint8_t> host_stage; // the same for host-only objects (mijpeg_prepare_batch_host)
  size_t stage_cap = 0;
  // batches (mijpeg_decode_batch_device): one parsed decoder per stream, frame 0's info with the batch's worst range
  std::vector<std::unique_ptr<HostDecoder>> batch_hosts;
  mijpeg_info batch_info{};
  int batch_frames = 0;
  // a submitted batch whose device work has not been waited for yet (mijpeg_submit_batch_device)
  int pend_n = 0;
  const uint32_t *pend_status = nullptr;
  int pend_walk_round = 0;                       // > 0: the batch went through the device walk with this many rounds, unchecked
  const uint32_t *pend_walk_flags = nullptr;     // "something changed" per round (pinned)
  const uint32_t *pend_walk_status = nullptr;    // per image (pinned)
  std::chrono::steady_clock::time_point pend_t0;
  // batches whose images bring different quantisation tables: [frames][4][64] deltas per component, on the device
  uint16_t *batch_quant_dev = nullptr;
  size_t batch_quant_cap = 0;
  bool batch_own_tables = false;
  std::vector<uint16_t> batch_quant_host;
  // MIJPEG_FLAG_SPECULATIVE: the reconstruction of a submitted batch was launched on an ASSUMED range check (spec_assumed:
  // what the last batch of this shape reported, rounded up to the kernel selection's next gate) behind the Huffman kernel,
  // without the host waiting for what that kernel reports; finish_batch validates and launches again where the assumption
  // did not hold (settle_speculation)
  bool spec_active = false, spec_redone = false;
  void *spec_dst = nullptr;
  int64_t spec_frame_stride = 0, spec_row_stride = 0;
  uint32_t spec_flags = 0;
  int32_t spec_assumed[MIJPEG_MAX_COMPONENTS] = {0, 0, 0, 0};
  int64_t spec_launched = 0, spec_redone_count = 0; // diagnostics (mijpeg_batch_speculation)
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t chain_ev = nullptr; // mijpeg_stream_wait
  hipEvent_t ms_ready = nullptr, ms_done = nullptr; // device_entropy_multiscan: the second frame's stream
  hipStream_t ms_stream = nullptr;
  int err_code = 0;
  std::string err_msg;
  double timing[4] = {0, 0, 0, 0};
};

static int set_error(mijpeg_decoder *d, int code, const std::string &msg)
{
  d->err_code = code;
  d->err_msg = msg;
  return code;
}

static int hip_fail(mijpeg_decoder *d, hipError_t e, const char *what)
{
  return set_error(d, MIJPEG_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// The boundary lets no C++ exception through (the reference turns everything into an error code at JPEG::Read /
// DisplayRectangle, interface/jpeg.cpp:205-220, tools/environment.hpp:752-784): every extern "C" body is a function-try-block
// whose handler lands here.  Out of memory is the reference's JPGERR_OUT_OF_MEMORY; anything else is a defect of this library
// and reported as such, not as a verdict on the stream.
static int boundary_catch(mijpeg_decoder *d, const char *where) noexcept
{
  int code = MIJPEG_ERR_PHASE_ERROR;
  const char *what = "unexpected exception";
  char buf[160];
  try {
    throw;
  } catch (const std::bad_alloc &) {
    code = MIJPEG_ERR_OUT_OF_MEMORY;
    what = "out of memory";
  } catch (const std::length_error &) { // (a container asked for more than max_size: memory all the same)
    code = MIJPEG_ERR_OUT_OF_MEMORY;
    what = "out of memory (container size)";
  } catch (const std::exception &e) {
    snprintf(buf, sizeof(buf), "%s", e.what());
    what = buf;
  } catch (...) {
  }
  if (d) {
    d->err_code = code;
    try {
      d->err_msg = std::string(where) + ": " + what;
    } catch (...) {
      d->err_msg.clear(); // (no memory for the message either: the code stands)
    }
  }
  return code;
}

#define HIP_TRY(d, call)                                  \
  do {                                                    \
    hipError_t e_ = (call);                               \
    if (e_ != hipSuccess) return hip_fail(d, e_, #call);  \
  } while (0)

// The large buffers of destroyed decoder objects -- pinned coefficient store and frame, their device mirrors -- wait here for
// the next object on the same device: a client that constructs a JPEG object per picture (cmd/reconstruct.cpp does) would
// otherwise pin ~200 MB of pages per 8K picture, 17 ms of the 24 ms such a decode took from Construct to Destruct.  At most
// four buffers per kind and 2 GiB in all are kept; a request takes the smallest buffer that fits and is at most twice as large.
namespace {
struct BufferCache {
  struct Entry { void *p; size_t bytes; int device; bool pinned; };
  std::mutex m;
  std::vector<Entry> kept;
  void *take(int device, bool pinned, size_t bytes, size_t *got)
  {
    std::lock_guard<std::mutex> lock(m);
    int best = -1;
    for (int i = 0; i < (int)kept.size(); i++)
      if (kept[(size_t)i].device == device && kept[(size_t)i].pinned == pinned && kept[(size_t)i].bytes >= bytes && kept[(size_t)i].bytes <= 2 * bytes &&
          (best < 0 || kept[(size_t)i].bytes < kept[(size_t)best].bytes))
        best = i;
    if (best < 0) return nullptr;
    void *p = kept[(size_t)best].p;
    *got = kept[(size_t)best].bytes;
    kept.erase(kept.begin() + best);
    return p;
  }
  // would give() keep a buffer like this one right now?
  bool has_room(int device, bool pinned, size_t bytes)
  {
    static const bool off = getenv("MIJPEG_NO_BUFFER_CACHE") != nullptr; // A-B measurements
    if (off || bytes < ((size_t)1 << 20)) return false;
    std::lock_guard<std::mutex> lock(m);
    return room_locked(device, pinned, bytes);
  }
  // false: not kept, the caller frees it
  bool give(int device, bool pinned, void *p, size_t bytes)
  {
    std::lock_guard<std::mutex> lock(m);
    if (!room_locked(device, pinned, bytes)) return false;
    kept.push_back(Entry{p, bytes, device, pinned});
    return true;
  }
  bool room_locked(int device, bool pinned, size_t bytes) const
  {
    size_t total = bytes, same = 0;
    for (const Entry &e : kept) {
      total += e.bytes;
      same += e.device == device && e.pinned == pinned;
    }
    return same < 4 && total <= limit_bytes();
  }
  // MIJPEG_BUFFER_CACHE_MB: what the cache may hold in all (default 2048, 0 = keep nothing)
  static size_t limit_bytes()
  {
    static const size_t lim = [] {
      const char *e = getenv("MIJPEG_BUFFER_CACHE_MB");
      return e ? (size_t)strtoull(e, nullptr, 10) << 20 : (size_t)2 << 30;
    }();
    return lim;
  }
  // mijpeg_trim_cache: hand everything back to the runtime
  size_t trim()
  {
    std::vector<Entry> gone;
    {
      std::lock_guard<std::mutex> lock(m);
      gone.swap(kept);
    }
    size_t bytes = 0;
    for (const Entry &e : gone) {
      bytes += e.bytes;
      if (e.pinned) (void)hipHostFree(e.p);
      else {
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(e.device);
        (void)hipFree(e.p);
        (void)hipSetDevice(cur);
      }
    }
    return bytes;
  }
};
BufferCache &buffer_cache()
{
  static BufferCache *c = new BufferCache; // (never destroyed: the HIP runtime may be gone when static destructors run)
  return *c;
}
} // namespace
static void release_big(int device, bool pinned, void *p, size_t bytes)
{
  if (!p) return;
  // hipFree / hipHostFree wait for the device before they take the memory away, and the buffers were handed to clients
  // (mijpeg_device_coefficients, mijpeg_batch: kernels on the client's own streams may still read them).  A buffer that changes
  // hands through the cache instead gets the same guarantee: nothing on the device is in flight when the next owner writes it.
  // Only a buffer that actually enters the cache needs it spelled out (and only its own device has to be idle): growing a
  // workspace in the middle of a pipeline must not stall every stream of the process for a buffer that is freed anyway.
  if (buffer_cache().has_room(device, pinned, bytes)) {
    if (device >= 0) {
      int cur = -1;
      (void)hipGetDevice(&cur);
      if (cur != device) (void)hipSetDevice(device);
      (void)hipDeviceSynchronize();
      if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
    if (buffer_cache().give(device, pinned, p, bytes)) return; // (another thread may have filled the room meanwhile: freed then)
  }
  if (pinned) (void)hipHostFree(p);
  else (void)hipFree(p);
}
// everything this object has enqueued is done (before one of its buffers changes hands while the object lives on: rare, a
// buffer only grows when a larger picture arrives)
static void quiesce(mijpeg_decoder *d)
{
  if (d->device < 0) return;
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->copy_stream) (void)hipStreamSynchronize(d->copy_stream);
  if (d->ms_stream) (void)hipStreamSynchronize(d->ms_stream);
}

// A batch that was submitted (mijpeg_submit_batch_device) and not waited for still reads the pinned staging buffers
// (ent_host, stage_host, status words) from its asynchronous uploads: every entry point that rewrites them settles it first.
static int settle_pending(mijpeg_decoder *d)
{
  if (!d->pend_n) return MIJPEG_OK;
  d->pend_n = 0;
  if (d->device >= 0) {
    HIP_TRY(d, hipSetDevice(d->device));
    HIP_TRY(d, hipStreamSynchronize(d->stream));
    if (d->copy_stream) HIP_TRY(d, hipStreamSynchronize(d->copy_stream));
  }
  return MIJPEG_OK;
}

extern "C" {

const char *mijpeg_version(void) { return "libjpeg_amd/mijpeg 0.1 (gfx950)"; }

int mijpeg_default_threads(void) { return default_threads(); }

size_t mijpeg_trim_cache(void)
try {
  return buffer_cache().trim();
} catch (...) { (void)boundary_catch(nullptr, "mijpeg_trim_cache"); return 0; }

int mijpeg_create(mijpeg_decoder **out, int device)
try {
  if (!out) return MIJPEG_ERR_INVALID_PARAMETER;
  *out = nullptr;
  mijpeg_decoder *d = new (std::nothrow) mijpeg_decoder();
  if (!d) return MIJPEG_ERR_OUT_OF_MEMORY;
  d->device = device;
  if (device >= 0) {
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&d->ev0);
    if (e == hipSuccess) e = hipEventCreate(&d->ev1);
    if (e != hipSuccess) {
      delete d;
      return MIJPEG_ERR_DEVICE;
    }
  }
  *out = d;
  return MIJPEG_OK;
} catch (...) { return boundary_catch(nullptr, "mijpeg_create"); }

void mijpeg_destroy(mijpeg_decoder *d)
try {
  if (!d) return;
  if (d->device >= 0) {
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->copy_stream) (void)hipStreamSynchronize(d->copy_stream); // (the buffers below may go to another object)
    if (d->ms_stream) (void)hipStreamSynchronize(d->ms_stream);
    release_big(d->device, true, d->coef_host, d->coef_host_cap * sizeof(int16_t));
    release_big(d->device, true, d->img_host, d->img_host_cap);
    release_big(d->device, false, d->coef_dev, d->coef_dev_cap * sizeof(int16_t));
    release_big(d->device, false, d->img_dev, d->img_dev_cap);
    if (d->ws_dev) (void)hipFree(d->ws_dev);
    if (d->batch_quant_dev) (void)hipFree(d->batch_quant_dev);
    if (d->ent_dev) (void)hipFree(d->ent_dev);
    if (d->ent_host) (void)hipHostFree(d->ent_host);
    if (d->stage_host) (void)hipHostFree(d->stage_host);
    if (d->walk_dev) (void)hipFree(d->walk_dev);
    if (d->xt_helper) mijpeg_destroy(d->xt_helper);
    if (d->enc_dev) (void)hipFree(d->enc_dev);
    for (int k = 0; k < 2; k++) {
      if (d->henc_dev[k]) (void)hipFree(d->henc_dev[k]);
      if (d->henc_out_dev[k]) (void)hipFree(d->henc_out_dev[k]);
    }
    if (d->henc_host) (void)hipHostFree(d->henc_host);
    if (d->walk_host) (void)hipHostFree(d->walk_host);
    if (d->req_dev) (void)hipFree(d->req_dev);
    if (d->req_host) (void)hipHostFree(d->req_host);
    if (d->rowmap_dev) (void)hipFree(d->rowmap_dev);
    if (d->ent_free) (void)hipEventDestroy(d->ent_free);
    for (hipEvent_t e : d->copy_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->band_events) (void)hipEventDestroy(e);
    if (d->copy_stream) (void)hipStreamDestroy(d->copy_stream);
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    if (d->chain_ev) (void)hipEventDestroy(d->chain_ev);
    if (d->ms_ready) (void)hipEventDestroy(d->ms_ready);
    if (d->ms_done) (void)hipEventDestroy(d->ms_done);
    if (d->ms_stream) (void)hipStreamDestroy(d->ms_stream);
    if (d->stream) (void)hipStreamDestroy(d->stream);
  } else {
    free(d->coef_host);
  }
  if (d->alpha) mijpeg_destroy(d->alpha);
  delete d;
} catch (...) { (void)boundary_catch(d, "mijpeg_destroy"); }

int mijpeg_set_input(mijpeg_decoder *d, const uint8_t *data, size_t size)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (!data) return set_error(d, MIJPEG_ERR_STREAM_EMPTY, "empty input stream");
  // a stream of no bytes: the reference's first GetWord meets the end of file, which is its SOI error (codestream/decoder.cpp:92-96)
  if (!size) return set_error(d, MIJPEG_ERR_MALFORMED_STREAM, "stream does not contain a JPEG file, SOI marker missing");
  d->data = data;
  d->size = size;
  d->parsed = d->decoded = d->uploaded = d->img_valid = d->model_valid = false;
  d->parse_fresh = false;
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_set_input"); }

int mijpeg_read_header(mijpeg_decoder *d, mijpeg_info *info)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (!d->data) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no input stream has been set");
  d->parse_fresh = false;
  const int rc = d->host.parse(d->data, d->size, true);
  if (rc) return set_error(d, rc, d->host.error.message);
  if (info) *info = d->host.info;
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_read_header"); }

static int ensure_dev(mijpeg_decoder *d, void **ptr, size_t *cap, size_t bytes);

static int ensure_coef_store(mijpeg_decoder *d, size_t count, bool need_host = true)
{
  if (need_host && d->coef_host_cap < count) {
    size_t cap = count;
    if (d->device >= 0) {
      quiesce(d); // (the buffer may go to another object: nothing of this one may still read or write it)
      release_big(d->device, true, d->coef_host, d->coef_host_cap * sizeof(int16_t));
      d->coef_host = nullptr;
      d->coef_host_cap = 0;
      size_t got = 0;
      if (void *p = buffer_cache().take(d->device, true, count * sizeof(int16_t), &got)) {
        d->coef_host = (int16_t *)p;
        cap = got / sizeof(int16_t);
      } else HIP_TRY(d, hipHostMalloc((void **)&d->coef_host, count * sizeof(int16_t), hipHostMallocDefault));
    } else {
      free(d->coef_host);
      d->coef_host = (int16_t *)malloc(count * sizeof(int16_t));
      if (!d->coef_host) return set_error(d, MIJPEG_ERR_OUT_OF_MEMORY, "out of memory for the coefficient store");
    }
    d->coef_host_cap = cap;
  }
  if (d->device >= 0 && d->coef_dev_cap < count) {
    quiesce(d);
    release_big(d->device, false, d->coef_dev, d->coef_dev_cap * sizeof(int16_t));
    d->coef_dev = nullptr;
    d->coef_dev_cap = 0;
    size_t got = 0;
    if (void *p = buffer_cache().take(d->device, false, count * sizeof(int16_t), &got)) {
      d->coef_dev = (int16_t *)p;
      d->coef_dev_cap = got / sizeof(int16_t);
    } else {
      HIP_TRY(d, hipMalloc((void **)&d->coef_dev, count * sizeof(int16_t)));
      d->coef_dev_cap = count;
    }
  }
  return MIJPEG_OK;
}

// The alpha channel of a JPEG XT file: the reference turns to the ALFA box behind the legacy codestream's EOI (and the residual
// codestream), inside JPEG::Read (Image::ParseTrailer, codestream/image.cpp:1430-1460): what is wrong with it fails the read,
// whether or not the client will ask for alpha.  Here a decoder object of its own -- same device, the file's boxes under the
// names an image's decoder looks for (HostDecoder::alpha_boxes) -- decodes it right behind the picture's codestreams.
static int decode_alpha_channel(mijpeg_decoder *d, int threads)
{
  d->alpha_ready = false;
  d->alpha_refusal = 0;
  if (!d->host.has_alpha()) return MIJPEG_OK;
  const uint8_t *p = nullptr;
  size_t n = 0;
  if (!d->host.alpha_stream(&p, &n)) return MIJPEG_OK;
  if (!d->alpha && mijpeg_create(&d->alpha, d->device) != MIJPEG_OK)
    return set_error(d, MIJPEG_ERR_OUT_OF_MEMORY, "no decoder object for the alpha channel");
  d->alpha_data.assign(p, p + n);
  d->alpha->host.preset_boxes(d->host.alpha_boxes());
  int rc = n ? mijpeg_set_input(d->alpha, d->alpha_data.data(), n)
             : set_error(d->alpha, MIJPEG_ERR_MALFORMED_STREAM, "Alpha channel codestream is invalid, SOI marker missing.");
  if (!rc) {
    // Image::ParseAlphaChannel compares the dimensions right behind the alpha FRAME HEADER (codestream/image.cpp:1366-1380), before
    // any of its scans is looked at (a frame header whose width byte is damaged: -1038, whatever its entropy coded data would do
    // to a frame of that size).  A height that arrives in a DNL marker still says 0 there.
    d->alpha->host.parse(d->alpha_data.data(), n, true); // (headers only; what it returns is the decode's to report)
    const mijpeg_info &a = d->alpha->host.info, &f = d->host.info;
    if (a.width > 0 && (a.width != f.width || a.height != f.height || a.dnl))
      rc = set_error(d->alpha, MIJPEG_ERR_MALFORMED_STREAM, "Malformed stream - residual image dimensions do not match the dimensions of the legacy image");
    else if (a.width > 0 && a.components != 1)
      rc = set_error(d->alpha, MIJPEG_ERR_MALFORMED_STREAM, "Malformed stream - the alpha channel may only consist of a single component");
  }
  if (!rc) rc = mijpeg_decode_coefficients(d->alpha, threads);
  if (!rc) {
    const mijpeg_info &a = d->alpha->host.info, &f = d->host.info;
    if (a.width != f.width || a.height != f.height) // codestream/image.cpp:1370-1380
      rc = set_error(d->alpha, MIJPEG_ERR_MALFORMED_STREAM, "Malformed stream - residual image dimensions do not match the dimensions of the legacy image");
    else if (a.components != 1)
      rc = set_error(d->alpha, MIJPEG_ERR_MALFORMED_STREAM, "Malformed stream - the alpha channel may only consist of a single component");
  }
  d->alpha_refusal = 0;
  if (rc == MIJPEG_ERR_OPERATION_UNIMPLEMENTED) {
    // (what this path declines does not fail the read -- but what stops the alpha image's codestreams does, and the reference has
    // read them whatever their specification says)
    const int v = d->alpha->host.declined_verdict();
    if (v && v != MIJPEG_ERR_OPERATION_UNIMPLEMENTED) rc = set_error(d->alpha, v, d->alpha->host.error.message.c_str());
  }
  if (rc) {
    const char *m = nullptr;
    mijpeg_last_error(d->alpha, &m);
    // What the alpha image's colour transformer would refuse (a table that does not exist ...) the reference only finds when
    // alpha pixels are asked for (Tables::ColorTrafoOf at the first request); what this path declines (-1034) is no reason to
    // withhold the picture either: the file reads, mijpeg_alpha_channel reports why there is no alpha.
    if (rc == MIJPEG_ERR_OPERATION_UNIMPLEMENTED || d->alpha->host.transformer_refused()) {
      d->alpha_refusal = rc;
      d->alpha_refusal_msg = m ? m : "";
      return MIJPEG_OK;
    }
    d->decoded = false; // the read has failed: no picture either (JPEG::Read returns false)
    return set_error(d, rc, m ? m : "the alpha channel does not decode");
  }
  d->alpha_ready = true;
  return MIJPEG_OK;
}

int mijpeg_decode_coefficients(mijpeg_decoder *d, int threads)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (!d->data) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no input stream has been set");
  if (d->device >= 0) HIP_TRY(d, hipSetDevice(d->device));
  if (const int prc = settle_pending(d)) return prc;
  d->batch_frames = 0;
  static const bool trace = getenv("MIJPEG_READ_TIMES") != nullptr; // diagnostics: where a read spends its time
  const auto t_begin = std::chrono::steady_clock::now();
  const bool parsed_already = d->parse_fresh; // (by mijpeg_decode_coefficients_device a moment ago)
  d->parse_fresh = false;
  int rc = parsed_already ? 0 : d->host.parse(d->data, d->size, false);
  if (rc) return set_error(d, rc, d->host.error.message);
  d->parsed = true;
  const mijpeg_info &f = d->host.info;
  const auto t_parsed = std::chrono::steady_clock::now();
  rc = ensure_coef_store(d, (size_t)f.coef_count);
  if (rc) return rc;
  const auto t_store = std::chrono::steady_clock::now();
  d->img_valid = d->model_valid = false;
  d->uploaded = false;
  d->host_planes_stale = false;
  d->batch_frames = 0;

  hipError_t copy_err = hipSuccess;
  std::function<void(int, int)> cb;
  if (d->device >= 0) {
    // stream finished MCU-row bands to the device while the workers decode the rest
    (void)hipEventRecord(d->ev0, d->stream);
    cb = [&](int r0, int r1) {
      for (int c = 0; c < f.components; c++) {
        const size_t row = (size_t)f.blocks_w[c] * 64 * f.vsamp[c]; // int16 per MCU row of this component
        const size_t off = (size_t)f.coef_offset[c] + row * r0, cnt = row * (r1 - r0);
        hipError_t e = hipMemcpyAsync(d->coef_dev + off, d->coef_host + off, cnt * sizeof(int16_t), hipMemcpyHostToDevice, d->stream);
        if (e != hipSuccess) copy_err = e;
      }
    };
  }
  // ... and the residual planes of a JPEG XT frame with hidden bits as the last refinement window finishes them (a worker thread
  // calls: its own device binding, its own error slot)
  std::atomic<int> rcopy_err{(int)hipSuccess};
  if (d->device >= 0 && d->host.residual()) {
    d->host.set_residual_rows_callback([&, d](int c, int y0, int y1) {
      (void)hipSetDevice(d->device);
      const mijpeg_xt_params &x = d->host.xt;
      const size_t row = (size_t)x.residual.blocks_w[c] * 64 * (x.residual_wide ? 2 : 1); // int16 units per block row
      const size_t off = (size_t)x.residual.coef_offset[c] + row * (size_t)y0;
      const hipError_t e = hipMemcpyAsync(d->coef_dev + off, d->coef_host + off, row * (size_t)(y1 - y0) * sizeof(int16_t), hipMemcpyHostToDevice, d->stream);
      if (e != hipSuccess) rcopy_err.store((int)e);
    });
  }
  rc = d->host.decode(d->coef_host, threads, cb);
  d->host.set_residual_rows_callback(nullptr);
  if (rcopy_err.load() != (int)hipSuccess && copy_err == hipSuccess) copy_err = (hipError_t)rcopy_err.load();
  d->timing[0] = d->host.huffman_seconds;
  if (trace) {
    const auto ms = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count() * 1e3; };
    fprintf(stderr, "read: parse %.2f ms, coefficient store %.2f ms, decode %.2f ms (entropy decoders %.2f)\n", ms(t_begin, t_parsed), ms(t_parsed, t_store),
            ms(t_store, std::chrono::steady_clock::now()), d->host.huffman_seconds * 1e3);
  }
  if (rc == MIJPEG_ERR_OVERFLOW_PARAMETER && !d->host.is_xt()) {
    // A coefficient beyond the 16-bit store -- only damaged streams get there: a DC prediction that runs away, a point
    // transform on garbage.  The reference keeps LONG coefficients and reconstructs what they hold; so does this frame,
    // in int32 planes (info.coef_wide) that the unfused kernels transform with the reference's 32-bit arithmetic.
    if (d->device >= 0) HIP_TRY(d, hipStreamSynchronize(d->stream)); // band uploads of the first attempt read coef_host
    d->parse_fresh = false;
    rc = d->host.parse(d->data, d->size, false);
    if (rc) return set_error(d, rc, d->host.error.message);
    rc = ensure_coef_store(d, (size_t)f.coef_count * 2);
    if (rc) return rc;
    rc = d->host.decode_wide((int32_t *)d->coef_host, threads);
    d->timing[0] += d->host.huffman_seconds;
    // (a frame with hidden refinement scans has no 32-bit planes on this path: declined, not the stream's fault)
    if (rc == MIJPEG_ERR_OVERFLOW_PARAMETER && d->host.left_16bit_store())
      return set_error(d, MIJPEG_ERR_OPERATION_UNIMPLEMENTED, "frame with hidden refinement scans and coefficients beyond the 16-bit store (a damaged scan) is not on the accelerated path");
    if (rc) return set_error(d, rc, d->host.error.message);
    if (d->device >= 0 && copy_err == hipSuccess)
      copy_err = hipMemcpyAsync(d->coef_dev, d->coef_host, (size_t)f.coef_count * sizeof(int16_t), hipMemcpyHostToDevice, d->stream);
  }
  // (a JPEG XT frame whose damaged scans leave the 16-bit store: the reference goes on in LONG coefficients, this path has no
  // int32 planes for merged frames -- declined like every other subset it does not take, not reported as the stream's fault)
  if (rc == MIJPEG_ERR_OVERFLOW_PARAMETER && d->host.is_xt() && d->host.left_16bit_store())
    return set_error(d, MIJPEG_ERR_OPERATION_UNIMPLEMENTED, "JPEG XT frame with coefficients beyond the 16-bit store (a damaged scan) is not on the accelerated path");
  if (rc) return set_error(d, rc, d->host.error.message);
  if (d->device >= 0 && d->host.residual() && copy_err == hipSuccess) {
    // the residual codestream's planes sit behind the legacy planes in the same buffer
    const mijpeg_xt_params &x = d->host.xt;
    bool some = false;
    for (int c = 0; c < x.residual.components; c++) some |= d->host.residual_rows_reported(c) > 0;
    if (!some) {
      const size_t off = (size_t)x.residual.coef_offset[0], cnt = (size_t)f.coef_count - off;
      copy_err = hipMemcpyAsync(d->coef_dev + off, d->coef_host + off, cnt * sizeof(int16_t), hipMemcpyHostToDevice, d->stream);
    } else // (rows the decode has sent on their way already: the rest of each plane)
      for (int c = 0; c < x.residual.components && copy_err == hipSuccess; c++) {
        const int y0 = d->host.residual_rows_reported(c), y1 = x.residual.blocks_h[c];
        if (y0 >= y1) continue;
        const size_t row = (size_t)x.residual.blocks_w[c] * 64 * (x.residual_wide ? 2 : 1);
        const size_t off = (size_t)x.residual.coef_offset[c] + row * (size_t)y0;
        copy_err = hipMemcpyAsync(d->coef_dev + off, d->coef_host + off, row * (size_t)(y1 - y0) * sizeof(int16_t), hipMemcpyHostToDevice, d->stream);
      }
  }
  if (copy_err != hipSuccess) return hip_fail(d, copy_err, "hipMemcpyAsync(coefficients)");
  d->decoded = true;
  if (d->device >= 0) {
    (void)hipEventRecord(d->ev1, d->stream);
    d->uploaded = true;
  }
  return decode_alpha_channel(d, threads);
} catch (...) { return boundary_catch(d, "mijpeg_decode_coefficients"); }

int64_t mijpeg_unstuffed_scan(mijpeg_decoder *d, uint8_t *dst, size_t capacity, uint32_t *begin, size_t n_begin, size_t piece_bytes,
                              int32_t *n_intervals)
{
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (!d->parsed || d->host.scans.empty()) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no parsed stream");
  if (const int prc = settle_pending(d)) return prc;
  if (piece_bytes == 1 && dst) { // the other producer: the marker search writes the copy itself (what a batch's workers do)
    if (capacity < d->size + 64) return set_error(d, MIJPEG_ERR_INVALID_PARAMETER, "the sink needs the stream's size (+ 64 bytes of slack)");
    d->host.set_unstuff_sink(dst, d->size);
    d->parse_fresh = false;
    const int rc = d->host.parse(d->data, d->size, false);
    if (rc) return set_error(d, rc, d->host.error.message);
    if (d->host.scans.empty() || d->host.scans[0].unstuffed_at != dst) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "the marker search did not write the copy");
    if (n_intervals) *n_intervals = (int32_t)d->host.scans[0].interval_ubegin.size();
    for (size_t k = 0; k < d->host.scans[0].interval_ubegin.size() && k < n_begin && begin; k++) begin[k] = d->host.scans[0].interval_ubegin[k];
    return (int64_t)d->host.scans[0].unstuffed_size;
  }
  const std::vector<uint32_t> &b = d->host.scans[0].interval_ubegin;
  const size_t total = d->host.scans[0].unstuffed_size;
  if (n_intervals) *n_intervals = (int32_t)b.size();
  for (size_t k = 0; k < b.size() && k < n_begin && begin; k++) begin[k] = b[k];
  if (dst && capacity >= total) {
    std::vector<HostDecoder::UnstuffPiece> pieces;
    d->host.unstuff_pieces(0, piece_bytes ? piece_bytes : ((size_t)1 << 20), pieces);
    for (const auto &p : pieces) d->host.unstuff_piece(0, p, dst);
  }
  return (int64_t)total;
}

int64_t mijpeg_speculative_scans(int64_t *pieces)
{
  if (pieces) *pieces = g_speculative_pieces.load();
  return g_speculative_scans.load();
}

int mijpeg_device_walk_rounds(mijpeg_decoder *d) { return d ? d->walk_rounds : 0; }

int mijpeg_get_info(mijpeg_decoder *d, mijpeg_info *info)
try {
  if (!d || !info) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->batch_frames < 0) { // a submitted batch: the range check is known once its Huffman kernel is through
    const int rc = mijpeg_finish_batch_device(d);
    if (rc) return rc;
  }
  if (d->batch_frames > 0) { // frame shape of the batch, range check of its most demanding image
    *info = d->batch_info;
    return MIJPEG_OK;
  }
  if (!d->data) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no input stream has been set");
  *info = d->host.info;
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_get_info"); }

int mijpeg_get_xt_params(mijpeg_decoder *d, mijpeg_xt_params *xt)
try {
  if (!d || !xt) return MIJPEG_ERR_INVALID_PARAMETER;
  if (!d->data || !d->host.is_xt()) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "the loaded stream is not a JPEG XT stream");
  *xt = d->host.xt;
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_get_xt_params"); }

const int32_t *mijpeg_coefficients32(mijpeg_decoder *d, int component)
{
  if (!d || !d->decoded || component < 0 || component >= d->host.info.components || !d->host.info.coef_wide) return nullptr;
  return (const int32_t *)(d->coef_host + d->host.info.coef_offset[component]); // wide frames are decoded on the host
}

const int16_t *mijpeg_coefficients(mijpeg_decoder *d, int component)
{
  if (!d || !d->decoded || component < 0 || component >= d->host.info.components) return nullptr;
  if (d->host.info.coef_wide) {
    set_error(d, MIJPEG_ERR_OVERFLOW_PARAMETER, "the frame holds 32-bit coefficients (info.coef_wide): mijpeg_coefficients32");
    return nullptr;
  }
  if (d->host_planes_stale) { // decoded on the device: fetch once
    if (hipSetDevice(d->device) != hipSuccess) return nullptr;
    if (ensure_coef_store(d, (size_t)d->host.info.coef_count, true)) return nullptr;
    if (hipMemcpyAsync(d->coef_host, d->coef_dev, (size_t)d->host.info.coef_count * sizeof(int16_t), hipMemcpyDeviceToHost, d->stream) != hipSuccess ||
        hipStreamSynchronize(d->stream) != hipSuccess)
      return nullptr;
    d->host_planes_stale = false;
  }
  return d->coef_host + d->host.info.coef_offset[component];
}

// ------------------------------------------------------------------------------------------------
// on-device entropy decoding
// ------------------------------------------------------------------------------------------------
// Why a parsed stream cannot be entropy-decoded on the device (nullptr: it can).  `xt_part`: the stream is one of the
// two codestreams of a JPEG XT profile C file (8-bit legacy or 12-bit residual frame without hidden refinement scans).
static const char *device_entropy_obstacle(const HostDecoder &h, size_t size, bool xt_part = false)
{
  const mijpeg_info &f = h.info;
  // one Huffman sequential scan over all components (or a single-component frame)
  if (h.needs_sequential())
    return "on-device entropy decoding: the stream is damaged; the host decoder walks it with the reference's resynchronisation (entropyparser.cpp:117-201)";
  if (f.progressive) return "on-device entropy decoding: progressive frames are decoded on the host";
  if (f.xt && !xt_part) return "on-device entropy decoding: not for this JPEG XT stream";
  if (!h.residual_merged()) return "on-device entropy decoding: the legacy codestream has no EOI marker (the host decoder decides what is merged)";
  // (a RESI box without a merging specification, a residual codestream header that does not match, tables looked up at the first
  // request: what the reference reports behind the legacy frame's decode -- HostDecoder::decode reports it, this path would not)
  if (h.verdict_pending()) return "on-device entropy decoding: the file's verdict is the host decoder's (residual codestream header / tables looked up at the first request)";
  if (f.precision != 8 && !(xt_part && f.precision == 12)) return "on-device entropy decoding: 8-bit frames (12-bit residual frames of JPEG XT) only";
  if (h.scans.size() != 1 || h.hidden_bits()) return "on-device entropy decoding: the frame has more than one scan";
  const Scan &s = h.scans[0];
  if (s.ncomp != f.components) return "on-device entropy decoding: the scan does not cover all components";
  if (size > 0xfffffff0ull) return "on-device entropy decoding: stream too long";
  for (int c = 0; c < f.components; c++)
    if (f.hsamp[c] > 4 || f.vsamp[c] > 4) return "on-device entropy decoding: MCUs of more than 4 x 4 blocks of a component are decoded on the host";
  return nullptr;
}

// Streams without restart markers: find their virtual restart intervals on the device.  Rounds of huffman_walk_kernel
// until the hand-over states between neighbouring subsequences stop changing, prefix sums over the subsequences
// (block numbers, DC predictors: huffman_walk_scan_kernel), and one EMIT walk that writes the interval tables the
// decode kernel reads.  The host only looks at the per-round "something changed" flags.
// `images_host` is the staging copy of the HuffImage array (first_interval = start of the image's interval entries).
// Where the device's copy of image i's entropy coded data goes inside the launch's stream buffer (and inside the pinned
// gathering area): a slot of the stream's own size -- known before the stream is parsed, so a batch's workers can write the
// copy while they search it for markers -- rounded to 16 bytes, plus the pad the kernels' prefetch may run into.
static size_t stream_slots(const size_t *sizes, int n, std::vector<size_t> &stream_off)
{
  stream_off.resize((size_t)n);
  size_t off = 0;
  for (int i = 0; i < n; i++) {
    stream_off[(size_t)i] = off;
    off += ((sizes[i] + 15) & ~(size_t)15) + HUFF_STREAM_PAD;
  }
  return off;
}

static int ensure_stage(mijpeg_decoder *d, size_t bytes)
{
  if (d->stage_cap >= bytes) return MIJPEG_OK;
  if (d->stage_host) (void)hipHostFree(d->stage_host);
  d->stage_host = nullptr;
  d->stage_cap = 0;
  HIP_TRY(d, hipHostMalloc((void **)&d->stage_host, bytes, hipHostMallocDefault));
  d->stage_cap = bytes;
  return MIJPEG_OK;
}

static int device_walk_images(mijpeg_decoder *d, HostDecoder *const *hosts, int n, const std::vector<int> &dwalk, const HuffScanArgs &scan,
                              const HuffImage *images_dev, uint32_t *ibegin_dev, uint8_t *iskip_dev, int16_t *ipred_dev,
                              const HuffImage *images_host, const std::vector<size_t> &usize, bool defer = false)
{
  const mijpeg_info &f0 = hosts[0]->info;
  const Scan &s0 = hosts[0]->scans[0];
  HuffWalkArgs w;
  memset(&w, 0, sizeof(w));
  w.ncomp = s0.ncomp;
  int B = 0;
  for (int k = 0; k < s0.ncomp; k++) {
    const int c = s0.sc[k].comp;
    w.hs[k] = s0.ncomp > 1 ? f0.hsamp[c] : 1;
    w.vs[k] = s0.ncomp > 1 ? f0.vsamp[c] : 1;
    B += w.hs[k] * w.vs[k];
  }
  if (B > 64) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "too many blocks per MCU for the device walk");
  w.nblk_mcu = B;
  w.ntables = scan.ntables;
  // subsequence size.  One image: the launch is latency-bound, its serial part is (distance the decoder needs to
  // synchronise + two subsequences), so small ones.  Batches are throughput-bound and every round re-walks whole
  // subsequences, so fewer rounds over larger ones (measured on 1, 4 and 16 8K frames: 128, 256, 512 bytes win).
  size_t longest = 0, all_bytes = 0;
  for (int i = 0; i < n; i++)
    if (dwalk[(size_t)i]) {
      const size_t len = usize[(size_t)i]; // (the device's copy: entropy coded data without the byte stuffing)
      longest = std::max(longest, len);
      all_bytes += len;
    }
  uint32_t sub_bytes = all_bytes <= ((size_t)8 << 20) ? 128 : all_bytes <= ((size_t)32 << 20) ? 256 : 512;
  while (sub_bytes < 1024 && longest / sub_bytes > ((size_t)1 << 20)) sub_bytes <<= 1; // bounds the prefix-sum tiles
  if (const char *e = getenv("MIJPEG_WALK_SUB")) sub_bytes = (uint32_t)std::max(32, std::min(4096, atoi(e))); // experiments
  w.sub_bytes = sub_bytes;
  // per image: its subsequences; per workgroup: image and first subsequence
  std::vector<uint32_t> img_sub0((size_t)n, 0), img_nsub((size_t)n, 0), img_e0((size_t)n, 0), img_e1((size_t)n, 0), img_int0((size_t)n, 0);
  uint32_t nsub_total = 0;
  for (int i = 0; i < n; i++) {
    const Scan &s = hosts[i]->scans[0];
    (void)s;
    img_e0[(size_t)i] = 0;
    img_e1[(size_t)i] = (uint32_t)usize[(size_t)i];
    img_int0[(size_t)i] = images_host[i].first_interval;
    img_sub0[(size_t)i] = nsub_total;
    if (dwalk[(size_t)i]) {
      img_nsub[(size_t)i] = (uint32_t)((usize[(size_t)i] + sub_bytes - 1) / sub_bytes);
      nsub_total += img_nsub[(size_t)i];
    }
  }
  w.lanes = 64;
  while (w.lanes > 1 && nsub_total / (uint32_t)w.lanes < 2048) w.lanes >>= 1;
  if (const char *e = getenv("MIJPEG_WALK_LANES")) w.lanes = std::max(1, std::min(64, atoi(e))); // experiments (power of two)
  w.waves_per_group = 4;
  const uint32_t per_group = (uint32_t)(w.lanes * w.waves_per_group);
  std::vector<uint32_t> sub_image, sub_first;
  for (int i = 0; i < n; i++)
    for (uint32_t k = 0; k < img_nsub[(size_t)i]; k += per_group) { sub_image.push_back((uint32_t)i); sub_first.push_back(k); }
  w.n_groups = (int32_t)sub_image.size();
  // one device buffer: [per group: image, first][per image: sub0, nsub, e0, e1, int0][per subsequence: state, stamp,
  // nblocks, dcsum, first_block, first_pred][changed flag per round][status per image]
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  constexpr int MAX_ROUNDS = 48;
  const size_t G = sub_image.size(), S = nsub_total;
  size_t o = 0;
  const size_t o_simg = o; o = al(o + G * 4);
  const size_t o_sfirst = o; o = al(o + G * 4);
  const size_t o_isub0 = o; o = al(o + (size_t)n * 4);
  const size_t o_insub = o; o = al(o + (size_t)n * 4);
  const size_t o_e0 = o; o = al(o + (size_t)n * 4);
  const size_t o_e1 = o; o = al(o + (size_t)n * 4);
  const size_t o_int0 = o; o = al(o + (size_t)n * 4);
  const size_t o_state = o; o = al(o + S * 8);
  const size_t o_up_end = o; // up to here the host fills the buffer
  const size_t o_flags = o; o = al(o + (size_t)(MAX_ROUNDS + 1) * 4 + (size_t)n * 4); // changed[], walk_status[]
  const size_t o_stamp = o; o = al(o + S * 4);
  const size_t o_zero_end = o; // flags and stamps start out as zero
  const size_t o_nblk = o; o = al(o + S * 4);
  const size_t o_dcsum = o; o = al(o + S * 16);
  const size_t o_fblk = o; o = al(o + S * 4);
  const size_t o_fpred = o; o = al(o + S * 16);
  uint32_t most = 0;
  for (int i = 0; i < n; i++) most = std::max(most, img_nsub[(size_t)i]);
  const int tiles = (int)((most + HUFF_WALK_TILE - 1) / HUFF_WALK_TILE);
  if (tiles > HUFF_WALK_TILE) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "stream too long for the device walk");
  const size_t o_tiles = o; o = al(o + (size_t)n * tiles * HUFF_WALK_SUMS_BYTES);
  const size_t total = o;
  int rc = ensure_dev(d, (void **)&d->walk_dev, &d->walk_cap, total);
  if (rc) return rc;
  const size_t host_bytes = o_up_end + (o_stamp - o_flags);
  if (d->walk_host_cap < host_bytes) {
    if (d->walk_host) (void)hipHostFree(d->walk_host);
    d->walk_host = nullptr;
    d->walk_host_cap = 0;
    HIP_TRY(d, hipHostMalloc((void **)&d->walk_host, host_bytes, hipHostMallocDefault));
    d->walk_host_cap = host_bytes;
  }
  uint8_t *wh = d->walk_host, *wd = d->walk_dev;
  memcpy(wh + o_simg, sub_image.data(), G * 4);
  memcpy(wh + o_sfirst, sub_first.data(), G * 4);
  memcpy(wh + o_isub0, img_sub0.data(), (size_t)n * 4);
  memcpy(wh + o_insub, img_nsub.data(), (size_t)n * 4);
  memcpy(wh + o_e0, img_e0.data(), (size_t)n * 4);
  memcpy(wh + o_e1, img_e1.data(), (size_t)n * 4);
  memcpy(wh + o_int0, img_int0.data(), (size_t)n * 4);
  // the initial guess: every subsequence starts at its boundary (behind a stuffed zero if it falls on one) with the
  // first block of an MCU; for the first subsequence of an image that is no guess
  {
    // (positions in the device's copy, which has no byte stuffing: nothing of the stream is looked at here)
    uint64_t *st = (uint64_t *)(wh + o_state);
    for (int i = 0; i < n; i++)
      for (uint32_t k = 0; k < img_nsub[(size_t)i]; k++) st[img_sub0[(size_t)i] + k] = (uint64_t)k * sub_bytes;
  }
  HIP_TRY(d, hipMemcpyAsync(wd, wh, o_up_end, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(d, hipMemsetAsync(wd + o_flags, 0, o_zero_end - o_flags, d->stream));
  w.data = scan.data;
  w.images = images_dev;
  w.tables = scan.tables;
  w.sub_image = (const uint32_t *)(wd + o_simg);
  w.sub_first = (const uint32_t *)(wd + o_sfirst);
  w.img_sub0 = (const uint32_t *)(wd + o_isub0);
  w.img_nsub = (const uint32_t *)(wd + o_insub);
  w.img_e0 = (const uint32_t *)(wd + o_e0);
  w.img_e1 = (const uint32_t *)(wd + o_e1);
  w.img_int0 = (const uint32_t *)(wd + o_int0);
  w.state = (uint64_t *)(wd + o_state);
  w.stamp = (uint32_t *)(wd + o_stamp);
  w.changed = (uint32_t *)(wd + o_flags);
  w.walk_status = w.changed + MAX_ROUNDS + 1;
  w.nblocks = (uint32_t *)(wd + o_nblk);
  w.dcsum = (int32_t *)(wd + o_dcsum);
  w.first_block = (uint32_t *)(wd + o_fblk);
  w.first_pred = (int32_t *)(wd + o_fpred);
  w.tile_sums = (WalkSums *)(wd + o_tiles);
  w.tiles_per_image = tiles;
  w.ibegin = ibegin_dev;
  w.iskip = iskip_dev;
  w.ipred = ipred_dev;
  const int64_t total_blocks = (int64_t)s0.mcus_x * s0.mcus_y * B;
  w.total_blocks = (uint32_t)total_blocks;
  // rounds, launched back to back in bunches; between bunches the host looks at the flags: a round that changed no
  // hand-over state means the states are the fixed point (and the counts of the lanes' last walks belong to it)
  uint32_t *flags_host = (uint32_t *)(wh + o_up_end);
  static const int first_bunch = getenv("MIJPEG_WALK_ROUNDS") ? std::max(1, std::min(MAX_ROUNDS, atoi(getenv("MIJPEG_WALK_ROUNDS")))) : 8;
  int round = 0;
  if (defer) {
    // mijpeg_submit_batch_device: nobody looks at the flags between the rounds.  Enough rounds for the states to settle are
    // launched in one go -- a round in which no lane is dirty costs a few microseconds (its workgroups leave before they
    // load their tables) -- and whoever waits for the batch checks that the last one changed nothing (finish_batch).
    const int rounds = std::min(MAX_ROUNDS, sub_bytes >= 512 ? 16 : sub_bytes >= 256 ? 24 : 40);
    while (round < rounds) {
      w.round = (uint32_t)++round;
      if (launch_huffman_walk(w, false, d->stream)) return hip_fail(d, hipGetLastError(), "huffman_walk_kernel launch");
    }
    HIP_TRY(d, hipMemcpyAsync(flags_host, wd + o_flags, (size_t)(MAX_ROUNDS + 1) * 4, hipMemcpyDeviceToHost, d->stream));
    d->pend_walk_round = round;
    d->pend_walk_flags = flags_host;
  }
  for (; !defer;) {
    const int upto = round == 0 ? first_bunch : std::min(MAX_ROUNDS, round + 4);
    while (round < upto) {
      w.round = (uint32_t)++round;
      if (launch_huffman_walk(w, false, d->stream)) return hip_fail(d, hipGetLastError(), "huffman_walk_kernel launch");
    }
    HIP_TRY(d, hipMemcpyAsync(flags_host, wd + o_flags, (size_t)(MAX_ROUNDS + 1) * 4, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(d, hipStreamSynchronize(d->stream));
    if (!flags_host[round]) break;
    if (round == MAX_ROUNDS) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "speculative decoding did not settle");
  }
  d->walk_rounds = 1;
  for (int r = 1; r <= round && !defer; r++)
    if (flags_host[r]) d->walk_rounds = r + 1; // rounds that were needed: the last one that changed something, and one to see it
  if (launch_huffman_walk_scan(w, n, d->stream)) return hip_fail(d, hipGetLastError(), "huffman_walk_scan_kernel launch");
  // one interval size for the launch: the images share their geometry, hence their MCUs per virtual interval
  int per = 0;
  for (int i = 0; i < n; i++)
    if (dwalk[(size_t)i]) per = dwalk[(size_t)i];
  w.emit_every = (uint32_t)(per * B);
  if (launch_huffman_walk(w, true, d->stream)) return hip_fail(d, hipGetLastError(), "huffman_walk_kernel launch");
  d->walk_status_dev = w.walk_status;
  return MIJPEG_OK;
}

// What the Huffman kernel left in the status words of n images: errors, and per image the range check that selects the
// arithmetic flavour of the reconstruction (fast_arith / range_max).
static int evaluate_entropy_status(mijpeg_decoder *d, HostDecoder *const *hosts, int n, const uint32_t *status_host)
{
  for (int i = 0; i < n; i++) {
    const uint32_t *st = status_host + 8 * i;
    // A DC prediction that leaves the 16-bit store (only damaged streams get there): the host decoder keeps 32-bit planes
    if (st[0] == HUFF_ERR_OVERFLOW)
      return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "on-device entropy decoding: a DC coefficient leaves the 16 bit coefficient store (damaged stream); the host decoder keeps 32-bit coefficients for it");
    // Damaged entropy coded data: which error the reference reports (or whether it decodes on after a resynchronisation)
    // depends on its sequential walk; the host decoder restates that walk, the device decoder does not try to
    if (st[0]) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "the entropy coded data is damaged: the host decoder walks such streams like the reference does (DESIGN 4.0)");
    mijpeg_info &f = hosts[i]->info;
    f.fast_arith = 1;
    for (int c = 0; c < f.components; c++) {
      f.range_max[c] = (int32_t)std::min<uint32_t>(st[1 + c], 0x7fffffffu);
      if (f.range_max[c] >= 16384) f.fast_arith = 0;
    }
    if (f.precision != 8) f.fast_arith = 0; // (as HostDecoder::decode has it: the fast flavour is derived for 8-bit frames; 12-bit kernels gate on range_max)
  }
  return MIJPEG_OK;
}

// Second-level tables of a device Huffman table: every code longer than the direct table's ten bits, grouped by its first
// ten bits, in a 64-entry table indexed by the six bits that follow (entries as in the direct table: huff_dev_entry).  Codes
// whose prefix finds no table left keep the direct entry HUFF_DEV_SUB | HUFF_DEV_NO_SUB: the kernels walk the canonical arrays
// for those.
static void fill_second_level(HuffDevTable &dst, const HuffTable &h, int ac)
{
  int prefix_of[HUFF_DEV_SUBTABLES], used = 0;
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < h.counts[l - 1]; i++, code++, k++) {
      if (l <= HUFF_DEV_LOOKAHEAD || k >= 256) continue;
      if (code >= (1 << l)) return; // over-subscribed lengths: the host refuses such tables anyway
      const int prefix = code >> (l - HUFF_DEV_LOOKAHEAD), rest = l - HUFF_DEV_LOOKAHEAD; // 1..6 bits behind the prefix
      int t = 0;
      while (t < used && prefix_of[t] != prefix) t++;
      if (t == used) {
        if (used == HUFF_DEV_SUBTABLES) continue;
        prefix_of[used++] = prefix;
        dst.fast[prefix] = (uint16_t)(HUFF_DEV_SUB | t);
      }
      const uint16_t e = (uint16_t)huff_dev_entry(l, h.values[k], ac);
      const int first = (code & ((1 << rest) - 1)) << (6 - rest);
      for (int j = 0; j < (1 << (6 - rest)); j++) dst.sub[t][first + j] = e;
    }
    code <<= 1;
  }
}

// The host's decoder table in the device's form (huffman_dev.hpp): direct entries, second-level tables, the canonical arrays.
// mode: 0 DC, 1 AC of a sequential scan, 2 AC of a progressive / refinement scan (huff_dev_entry).
static void build_dev_table(HuffDevTable &dst, const HuffTable &src, int mode)
{
  memset(&dst, 0, sizeof(dst));
  // the host's direct table ((length << 8) | symbol, 0 = a longer code or none) in the device's entry format
  for (int x = 0; x < (1 << HUFF_DEV_LOOKAHEAD); x++) {
    const uint16_t he = src.fast[x];
    dst.fast[x] = he ? (uint16_t)huff_dev_entry(he >> 8, he & 0xffu, mode) : (uint16_t)(HUFF_DEV_SUB | HUFF_DEV_NO_SUB);
  }
  static const bool no_sub = getenv("MIJPEG_HUFF_NO_SUBTABLES") != nullptr; // A-B measurements: long codes walk the canonical arrays
  if (!no_sub) fill_second_level(dst, src, mode);
  memcpy(dst.maxcode, src.maxcode, sizeof(dst.maxcode));
  memcpy(dst.valoff, src.valoff, sizeof(dst.valoff));
  memcpy(dst.values, src.values, sizeof(dst.values));
}

// Entropy-decode n parsed images of identical frame geometry on the device, image i into coef_dev + i * frame_stride.
// infos[i] receives fast_arith / range_max.  Returns MIJPEG_OK, MIJPEG_ERR_NOT_AVAILABLE (nothing touched) or an error.
static int device_entropy_batch(mijpeg_decoder *d, HostDecoder *const *hosts, const uint8_t *const *datas, const size_t *sizes, int n,
                                int min_intervals, int16_t *coef_dev, int64_t frame_stride, bool xt_part = false, bool defer = false)
{
  const mijpeg_info &f0 = hosts[0]->info;
  const Scan &s0 = hosts[0]->scans[0];
  int64_t total_intervals = 0;
  std::vector<int64_t> nints((size_t)n);
  std::vector<std::unique_ptr<VirtualIntervals>> virt((size_t)n); // restart points planned by the host's walk ...
  std::vector<int> dwalk((size_t)n, 0);                           // ... or MCUs per virtual interval when the device walks
  d->walk_rounds = 0;
  const bool device_walk = !(getenv("MIJPEG_DEVICE_WALK") && atoi(getenv("MIJPEG_DEVICE_WALK")) == 0);
  const auto tb0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) {
    const char *why = device_entropy_obstacle(*hosts[i], sizes[i], xt_part);
    if (why) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, why);
    const mijpeg_info &f = hosts[i]->info;
    const Scan &s = hosts[i]->scans[0];
    if (f.width != f0.width || f.height != f0.height || f.components != f0.components || memcmp(f.hsamp, f0.hsamp, sizeof(f.hsamp)) ||
        memcmp(f.vsamp, f0.vsamp, sizeof(f.vsamp)))
      return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "the images of a batch must share width, height and sampling factors");
    // ... and what the one reconstruction launch applies to all of them: the colour transformation (an Adobe marker may
    // switch it off per image), the sample precision, being a JPEG XT stream or not
    if (f.ycbcr != f0.ycbcr || f.precision != f0.precision || f.xt != f0.xt)
      return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "the images of a batch must share colour transformation and precision");
    for (int k = 0; k < s.ncomp; k++)
      if (s.sc[k].comp != s0.sc[k].comp) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "the images of a batch must share the component order of their scan");
    const int64_t total_mcus = (int64_t)s.mcus_x * s.mcus_y;
    int64_t nint;
    if (s.restart_interval > 0) {
      nint = (total_mcus + s.restart_interval - 1) / s.restart_interval;
      if (nint > 0x7fffffff) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "too many restart intervals");
      if ((int64_t)s.interval_begin.size() < nint)
        return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "restart markers missing: the host decoder resynchronises like the reference (entropyparser.cpp:117-201)");
      const std::vector<uint8_t> &rst = hosts[i]->restart_codes(0);
      for (int64_t k = 0; k + 1 < nint; k++)
        if (rst[(size_t)k] != 0xd0 + (k & 7))
          return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "restart markers out of sequence: the host decoder resynchronises like the reference (entropyparser.cpp:117-201)");
    } else {
      // no restart markers: the host's self-synchronising walk finds exact restart points ("virtual intervals"),
      // about 16 K of them, and the device decodes from there
      const int per = (int)std::min<int64_t>(64, std::max<int64_t>(1, total_mcus / 16384));
      if (total_mcus < 256 || s.ecs_end - s.ecs_begin < 4096)
        return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "stream without restart markers is too small for speculative decoding");
      if (device_walk) { // the restart points are found on the device (huffman_walk_kernel); their number is known already
        dwalk[(size_t)i] = per;
        nint = (total_mcus + per - 1) / per;
      } else {
        virt[(size_t)i].reset(new VirtualIntervals());
        if (hosts[i]->plan_virtual_intervals(0, per, 0, *virt[(size_t)i]))
          return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "stream without restart markers did not lend itself to speculative decoding");
        nint = (int64_t)virt[(size_t)i]->byte_off.size();
      }
    }
    nints[(size_t)i] = nint;
    total_intervals += nint;
  }
  if (min_intervals <= 0) min_intervals = 2048; // below this the device runs mostly idle
  if (total_intervals < min_intervals || total_intervals > 0x7fffffff)
    return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "too few restart intervals to occupy the device");

  HuffScanArgs a;
  memset(&a, 0, sizeof(a));
  // decoding lanes per wave: about a thousand waves (one per SIMD) are what a small launch wants -- fewer lanes per wave
  // mean more waves that each issue the same instructions for less, fuller waves mean longer steps (the slowest lane's
  // block).  Measured with the bit-addressed reader and four-wave workgroups on one 8K 4:2:0 frame with 16200 intervals
  // (tools/gpu_huff_lanes.sh): 1 lane 0.81 ms, 2: 0.52, 4: 0.33, 8: 0.26, 16: 0.26, 32: 0.27.
  a.lanes = 64;
  while (a.lanes > 1 && total_intervals / a.lanes < 768) a.lanes >>= 1;
  if (const char *e = getenv("MIJPEG_HUFF_LANES")) { // tuning
    const int l = atoi(e);
    if (l >= 1 && l <= 64 && (l & (l - 1)) == 0) a.lanes = l;
  }
  // one wave per SIMD and workgroup: with two-wave workgroups (which round 1 chose for the LDS they leave to others) the same
  // eight waves per CU decode 37 % slower (0.60 against 0.38 ms per 32 4K frames; 3, 5, 6 waves: 0.55, 0.58, 0.48) -- the
  // waves of a workgroup go to the SIMDs in cyclic order, and only a multiple of four loads them evenly
  a.waves_per_group = 4;
  if (const char *e = getenv("MIJPEG_HUFF_WAVES")) a.waves_per_group = std::max(1, std::min(8, atoi(e))); // tuning
  const int per_group = a.lanes * a.waves_per_group; // intervals of one workgroup
  // Tables in LDS: components that bring the same Huffman code (Cb and Cr practically always do) share one copy -- the
  // workgroup's LDS footprint decides how many of them a CU holds.  The sharing pattern is that of image 0 and must hold
  // for every image of the launch; the device walk indexes its tables by component and keeps one per component.
  int tab_slot[MIJPEG_MAX_COMPONENTS][2];
  int ntab = 0;
  {
    bool walk_any = false;
    for (int i = 0; i < n; i++) walk_any |= dwalk[(size_t)i] > 0;
    bool share = !walk_any && !getenv("MIJPEG_HUFF_NO_TABLE_SHARING");
    for (int pass = 0; pass < 2; pass++) {
      ntab = 0;
      for (int k = 0; k < s0.ncomp; k++)
        for (int t = 0; t < 2; t++) {
          tab_slot[k][t] = -1;
          for (int j = 0; j < k && share && tab_slot[k][t] < 0; j++)
            if ((t ? s0.ac[k].same_code(s0.ac[j]) : s0.dc[k].same_code(s0.dc[j]))) tab_slot[k][t] = tab_slot[j][t];
          if (tab_slot[k][t] < 0) tab_slot[k][t] = ntab++;
        }
      if (!share) break;
      bool holds = true; // ... in every image?
      for (int i = 1; i < n && holds; i++) {
        const Scan &s = hosts[i]->scans[0];
        for (int k = 0; k < s.ncomp && holds; k++)
          for (int j = 0; j < k && holds; j++) {
            if (tab_slot[k][0] == tab_slot[j][0] && !s.dc[k].same_code(s.dc[j])) holds = false;
            if (tab_slot[k][1] == tab_slot[j][1] && !s.ac[k].same_code(s.ac[j])) holds = false;
          }
      }
      if (holds) break;
      share = false;
    }
  }
  const size_t table_blob = (size_t)ntab * sizeof(HuffDevTable) + sizeof(HuffDevAux);

  // device buffer: [streams, each padded][ibegin][iend][tables of every image][images][groups][status]
  // What travels to the device is the entropy coded data of every image WITHOUT its byte stuffing and without the markers,
  // one restart interval behind the other (HostDecoder::unstuff_piece; the marker search counted what leaves): the kernels
  // address it by plain bit positions (huffman.hip, DevBits).
  std::vector<size_t> usize((size_t)n);
  std::vector<size_t> stream_off;
  size_t off = stream_slots(sizes, n, stream_off);
  int64_t n_groups = 0;
  for (int i = 0; i < n; i++) {
    usize[(size_t)i] = hosts[i]->scans[0].unstuffed_size;
    if (usize[(size_t)i] >= ((size_t)1 << 28)) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "entropy coded segment too large for the device decoder's bit addresses");
    if (usize[(size_t)i] > sizes[i]) return set_error(d, MIJPEG_ERR_INVALID_PARAMETER, "entropy coded segment larger than its stream");
    if (!dwalk[(size_t)i] && !virt[(size_t)i] && (int64_t)hosts[i]->scans[0].interval_ubegin.size() < nints[(size_t)i])
      return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "restart intervals missing");
    n_groups += (nints[(size_t)i] + per_group - 1) / per_group;
  }
  if (off > 0xfffffff0ull || n_groups > 0x7fffffff) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "batch too large for one launch");
  const size_t stream_bytes = off;
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  bool any_virtual = false, any_dwalk = false;
  for (int i = 0; i < n; i++) {
    any_virtual |= virt[(size_t)i] != nullptr || dwalk[(size_t)i] > 0;
    any_dwalk |= dwalk[(size_t)i] > 0;
  }
  const size_t off_ib = stream_bytes, off_ie = off_ib + (size_t)total_intervals * 4;
  const size_t off_isk = off_ie + (size_t)total_intervals * 4, off_ipr = align16(off_isk + (any_virtual ? (size_t)total_intervals : 0));
  const size_t off_tab = align16(off_ipr + (any_virtual ? (size_t)total_intervals * 8 : 0));
  const size_t off_img = align16(off_tab + (size_t)n * table_blob), off_grp = align16(off_img + (size_t)n * sizeof(HuffImage));
  const size_t off_status = align16(off_grp + (size_t)n_groups * sizeof(HuffGroup)), status_bytes = (size_t)n * 32, total = off_status + status_bytes;
  int rc = ensure_dev(d, (void **)&d->ent_dev, &d->ent_cap, total);
  if (rc) return rc;
  const size_t host_part = off_status - stream_bytes; // everything between the streams and the status words goes through pinned staging
  if (d->ent_host_cap < host_part + status_bytes) {
    if (d->ent_host) (void)hipHostFree(d->ent_host);
    d->ent_host = nullptr;
    d->ent_host_cap = 0;
    HIP_TRY(d, hipHostMalloc((void **)&d->ent_host, host_part + status_bytes, hipHostMallocDefault));
    d->ent_host_cap = host_part + status_bytes;
  }
  uint8_t *hp = d->ent_host - stream_bytes; // hp + device offset = staging address
  uint32_t *ib = (uint32_t *)(hp + off_ib), *ie = (uint32_t *)(hp + off_ie);
  HuffImage *images = (HuffImage *)(hp + off_img);
  HuffGroup *groups = (HuffGroup *)(hp + off_grp);
  int64_t first = 0, g = 0;
  bool needs_clear = false;
  for (int i = 0; i < n; i++) {
    const mijpeg_info &f = hosts[i]->info;
    const Scan &s = hosts[i]->scans[0];
    const int64_t nint = nints[(size_t)i];
    if (dwalk[(size_t)i]) { // filled in by the EMIT walk on the device
      for (int64_t k = 0; k < nint; k++) { ib[first + k] = 0; ie[first + k] = (uint32_t)usize[(size_t)i]; }
      memset(hp + off_isk + first, 0, (size_t)nint);
      memset(hp + off_ipr + (size_t)first * 8, 0, (size_t)nint * 8);
    } else if (virt[(size_t)i]) {
      // the host's walk reports stream offsets: into the copy's (stuffed pairs in front of each, counted as the offsets go up)
      const VirtualIntervals &vi = *virt[(size_t)i];
      uint8_t *isk = hp + off_isk;
      int16_t *ipr = (int16_t *)(hp + off_ipr);
      const uint8_t *base = s.base ? s.base : datas[i];
      size_t at = s.ecs_begin, pairs = 0;
      for (int64_t k = 0; k < nint; k++) {
        const size_t pos = vi.byte_off[(size_t)k];
        while (at < pos) {
          const uint8_t *q = (const uint8_t *)memchr(base + at, 0xff, pos - at);
          if (!q) break;
          at = (size_t)(q - base);
          if (base[at + 1] == 0x00) { pairs++; at += 2; }
          else at++;
        }
        at = std::max(at, pos);
        ib[first + k] = (uint32_t)(pos - s.ecs_begin - pairs);
        ie[first + k] = (uint32_t)usize[(size_t)i];
        isk[first + k] = vi.bit_skip[(size_t)k];
        memcpy(ipr + (first + k) * 4, &vi.pred[(size_t)k * 4], 8);
      }
    } else {
      memcpy(ib + first, s.interval_ubegin.data(), (size_t)nint * 4);
      memcpy(ie + first, s.interval_uend.data(), (size_t)nint * 4);
      if (any_virtual) memset(hp + off_isk + first, 0, (size_t)nint);
    }
    HuffDevTable *tabs = (HuffDevTable *)(hp + off_tab + (size_t)i * table_blob);
    HuffDevAux *aux = (HuffDevAux *)(tabs + ntab);
    // images that bring the tables of the image in front of them (every frame of a camera or an encoder run does) share its
    // blob: nothing to build, and the workgroups of both read the same lines
    bool same_tables = i > 0;
    if (same_tables) {
      const mijpeg_info &fp = hosts[i - 1]->info;
      const Scan &sp = hosts[i - 1]->scans[0];
      for (int k = 0; k < s.ncomp && same_tables; k++) {
        const int c = s.sc[k].comp;
        same_tables = s.dc[k].same_code(sp.dc[k]) && s.ac[k].same_code(sp.ac[k]) &&
                      !memcmp(f.quant[f.quant_index[c]], fp.quant[fp.quant_index[c]], sizeof(f.quant[0]));
      }
    }
    memset(aux, 0, sizeof(*aux));
    for (int k = 0; k < s.ncomp && !same_tables; k++) {
      const HuffTable *src[2] = {&s.dc[k], &s.ac[k]};
      for (int t = 0; t < 2; t++) {
        build_dev_table(tabs[tab_slot[k][t]], *src[t], t);
      }
      const int c = s.sc[k].comp;
      const uint16_t *delta = f.quant[f.quant_index[c]];
      for (int z = 0; z < 80; z++) {
        const uint32_t pos = scan_order()[z];
        aux->zq[k][z] = ((uint32_t)delta[pos] << 16) | (pos * 2);
      }
    }
    HuffImage &im = images[i];
    im.stream_off = (uint32_t)stream_off[(size_t)i];
    im.first_interval = (uint32_t)first;
    im.n_intervals = (int32_t)nint;
    im.restart_interval = dwalk[(size_t)i] ? dwalk[(size_t)i] : virt[(size_t)i] ? virt[(size_t)i]->mcus_per_interval : s.restart_interval;
    im.virt = (virt[(size_t)i] || dwalk[(size_t)i]) ? 1u : 0u;
    im.reserved = 0;
    im.total_mcus = s.mcus_x * s.mcus_y;
    im.mcus_x = s.mcus_x;
    im.coef_base = (int64_t)i * frame_stride;
    im.table_off = same_tables ? images[i - 1].table_off : (uint32_t)((size_t)i * table_blob);
    im.status_off = (uint32_t)(i * 8);
    for (int64_t k = 0; k < nint; k += per_group) {
      groups[g].image = (uint32_t)i;
      groups[g].first_interval = (uint32_t)k;
      g++;
    }
    first += nint;
    // an interleaved scan writes every block of every plane; a single-component scan of a frame whose only component
    // has sampling factors > 1 leaves the MCU padding blocks untouched (they must read as zero)
    if (s.ncomp == 1 && (s.mcus_x != f.blocks_w[s.sc[0].comp] || s.mcus_y != f.blocks_h[s.sc[0].comp])) needs_clear = true;
  }
  for (int k = 0; k < s0.ncomp; k++) {
    const int c = s0.sc[k].comp;
    a.comp_of[k] = c;
    a.hs[k] = s0.ncomp > 1 ? f0.hsamp[c] : 1;
    a.vs[k] = s0.ncomp > 1 ? f0.vsamp[c] : 1;
    a.bw[k] = f0.blocks_w[c];
    a.coef_off[k] = f0.coef_offset[c];
    a.dc_tab[k] = tab_slot[k][0];
    a.ac_tab[k] = tab_slot[k][1];
  }
  a.data = d->ent_dev;
  a.ibegin = (const uint32_t *)(d->ent_dev + off_ib);
  a.iend = (const uint32_t *)(d->ent_dev + off_ie);
  a.iskip = d->ent_dev + off_isk;
  a.ipred = (const int16_t *)(d->ent_dev + off_ipr);
  a.images = (const HuffImage *)(d->ent_dev + off_img);
  a.groups = (const HuffGroup *)(d->ent_dev + off_grp);
  a.n_groups = (int32_t)n_groups;
  a.ncomp = s0.ncomp;
  a.ntables = ntab;
  a.tables = d->ent_dev + off_tab;
  a.coef = coef_dev;
  a.status = (uint32_t *)(d->ent_dev + off_status);
  const auto tb1 = std::chrono::steady_clock::now();
  static const bool trace_phases = getenv("MIJPEG_TRACE_SUBMIT") != nullptr; // diagnostics: host time of the steps below, on stderr
  auto mark = [&](const char *what) {
    if (trace_phases) fprintf(stderr, "[mijpeg] %-28s %8.3f ms\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tb1).count() * 1e3);
  };
  HIP_TRY(d, hipMemcpyAsync(d->ent_dev + stream_bytes, d->ent_host, host_part, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(d, hipMemsetAsync(d->ent_dev + off_status, 0, status_bytes, d->stream));
  if (needs_clear) HIP_TRY(d, hipMemsetAsync(coef_dev, 0, (size_t)n * (size_t)frame_stride * sizeof(int16_t), d->stream));
  const int repeat = getenv("MIJPEG_HUFF_REPEAT") ? atoi(getenv("MIJPEG_HUFF_REPEAT")) : 1; // experiments: steady-state kernel time
  // (a deferred batch always goes through the pinned gathering area: the caller's bytes are only read during the call)
  const bool small = !defer && (n == 1 || stream_bytes < ((size_t)8 << 20));
  {
    const int src = ensure_stage(d, stream_bytes);
    if (src) return src;
  }
  // the unstuffing gather of images [g0, g1) into the pinned area, spread over the pool: pieces of ~256 KiB of source
  // (images whose marker search wrote the copy already -- a batch's workers do, set_unstuff_sink -- have nothing left to do)
  auto gather = [&](int g0, int g1) {
    struct Job { int image; HostDecoder::UnstuffPiece piece; };
    std::vector<Job> jobs;
    std::vector<HostDecoder::UnstuffPiece> ps;
    for (int i = g0; i < g1; i++) {
      if (hosts[i]->scans[0].unstuffed_at == d->stage_host + stream_off[(size_t)i]) continue;
      ps.clear();
      hosts[i]->unstuff_pieces(0, (size_t)256 << 10, ps);
      for (const auto &p : ps) jobs.push_back(Job{i, p});
    }
    // (the sweep runs at about half of memcpy's rate: twice the workers the plain copy had)
    if (jobs.empty()) return;
    const int workers = std::max(1, std::min<int>((int)jobs.size(), std::min(default_threads(), 32)));
    parallel_for(workers, [&](int w) {
      for (size_t k = (size_t)w; k < jobs.size(); k += (size_t)workers) {
        const int i = jobs[k].image;
        hosts[i]->unstuff_piece(0, jobs[k].piece, d->stage_host + stream_off[(size_t)i]);
      }
    });
  };
  if (small) {
    gather(0, n);
    // (only what the copies occupy: a slot is as large as its stream, headers and all)
    for (int i = 0; i < n; i++)
      HIP_TRY(d, hipMemcpyAsync(d->ent_dev + stream_off[(size_t)i], d->stage_host + stream_off[(size_t)i], ((usize[(size_t)i] + 15) & ~(size_t)15) + HUFF_STREAM_PAD,
                                hipMemcpyHostToDevice, d->stream));
    if (any_dwalk) {
      const int wrc = device_walk_images(d, hosts, n, dwalk, a, (const HuffImage *)(d->ent_dev + off_img), (uint32_t *)(d->ent_dev + off_ib),
                                         d->ent_dev + off_isk, (int16_t *)(d->ent_dev + off_ipr), images, usize, defer);
      if (wrc) return wrc;
    }
    for (int r = 0; r < std::max(1, repeat); r++)
      if (launch_huffman_scan(a, d->stream)) return hip_fail(d, hipGetLastError(), "huffman_scan_kernel launch");
  } else {
    // large batches, in up to eight groups of images: the pool threads gather a group's streams into pinned memory, its
    // DMA runs on a copy stream while the next group is gathered and while the kernel decodes the previous one
    mark("staging buffer ready");
    if (!d->copy_stream) HIP_TRY(d, hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
    // Images per upload + launch.  A launch is latency-bound (the serial symbol chain of its longest restart interval,
    // ~0.3 ms) until it holds several waves per SIMD: ~128 K restart intervals; more, smaller launches only pay when the
    // batch is so large that the upload of one part hides behind the decode of another (profiles/r02/batch4k_*.txt:
    // 32 x 4K frames in one launch 0.80 ms, in eight launches of four 8 x 0.39 ms).
    const int64_t per_image = std::max<int64_t>(1, total_intervals / n);
    int groups_of = (int)std::max<int64_t>(std::max(4, (n + 7) / 8), (131072 + per_image - 1) / per_image);
    if (const char *e = getenv("MIJPEG_BATCH_GROUP")) groups_of = std::max(1, atoi(e)); // tuning
    const int ngroups = (n + groups_of - 1) / groups_of;
    while ((int)d->copy_events.size() < ngroups) {
      hipEvent_t e;
      HIP_TRY(d, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      d->copy_events.push_back(e);
    }
    // the copy stream must not overtake work that still reads the buffer from an earlier call: the Huffman kernels and the
    // status copy of the previous batch (ent_free, recorded behind them).  NOT everything on d->stream: the reconstruction
    // kernel of the previous batch does not touch this buffer, and an upload that waits for it leaves the link idle for the
    // length of that kernel in every round of a pipeline (profiles/r03/batch4k_timeline.txt)
    if (d->ent_free_valid) HIP_TRY(d, hipStreamWaitEvent(d->copy_stream, d->ent_free, 0));
    int64_t wg0 = 0;
    for (int gi = 0, g0 = 0; g0 < n; g0 += groups_of, gi++) {
      const int g1 = std::min(n, g0 + groups_of);
      gather(g0, g1);
      const size_t b0 = stream_off[(size_t)g0], b1 = g1 < n ? stream_off[(size_t)g1] : stream_bytes;
      HIP_TRY(d, hipMemcpyAsync(d->ent_dev + b0, d->stage_host + b0, b1 - b0, hipMemcpyHostToDevice, d->copy_stream));
      HIP_TRY(d, hipEventRecord(d->copy_events[(size_t)gi], d->copy_stream));
      HIP_TRY(d, hipStreamWaitEvent(d->stream, d->copy_events[(size_t)gi], 0));
      if (any_dwalk) continue; // the walk below covers all images at once
      int64_t wg1 = wg0;
      for (int i = g0; i < g1; i++) wg1 += (nints[(size_t)i] + per_group - 1) / per_group;
      HuffScanArgs part = a; // the workgroups of this group's images
      part.groups = a.groups + wg0;
      part.n_groups = (int32_t)(wg1 - wg0);
      for (int r = 0; r < std::max(1, repeat); r++)
        if (launch_huffman_scan(part, d->stream)) return hip_fail(d, hipGetLastError(), "huffman_scan_kernel launch");
      wg0 = wg1;
    }
    mark("groups gathered + enqueued");
    if (any_dwalk) {
      const int wrc = device_walk_images(d, hosts, n, dwalk, a, (const HuffImage *)(d->ent_dev + off_img), (uint32_t *)(d->ent_dev + off_ib),
                                         d->ent_dev + off_isk, (int16_t *)(d->ent_dev + off_ipr), images, usize, defer);
      if (wrc) return wrc;
      mark("device walk enqueued");
      for (int r = 0; r < std::max(1, repeat); r++)
        if (launch_huffman_scan(a, d->stream)) return hip_fail(d, hipGetLastError(), "huffman_scan_kernel launch");
    }
  }
  uint32_t *status_host = (uint32_t *)(d->ent_host + host_part);
  HIP_TRY(d, hipMemcpyAsync(status_host, d->ent_dev + off_status, status_bytes, hipMemcpyDeviceToHost, d->stream));
  uint32_t *walk_status_host = (uint32_t *)d->walk_host; // the walk's staging buffer is free again
  if (any_dwalk) HIP_TRY(d, hipMemcpyAsync(walk_status_host, d->walk_status_dev, (size_t)n * 4, hipMemcpyDeviceToHost, d->stream));
  if (!d->ent_free) HIP_TRY(d, hipEventCreateWithFlags(&d->ent_free, hipEventDisableTiming));
  HIP_TRY(d, hipEventRecord(d->ent_free, d->stream)); // from here on nothing enqueued so far reads the entropy buffers
  d->ent_free_valid = true;
  d->phase_prepare = std::chrono::duration<double>(tb1 - tb0).count(); // interval tables, Huffman tables
  mark("status copies enqueued");
  if (!any_dwalk) d->pend_walk_round = 0;
  if (defer) { // mijpeg_submit_batch_device: the caller waits later (finish_batch)
    d->pend_n = n;
    d->pend_status = status_host;
    d->pend_walk_status = any_dwalk ? walk_status_host : nullptr;
    d->pend_t0 = tb1;
    d->phase_device = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb1).count(); // so far: gathering + enqueueing
    return MIJPEG_OK;
  }
  HIP_TRY(d, hipStreamSynchronize(d->stream));
  if (any_dwalk)
    for (int i = 0; i < n; i++) {
      if (walk_status_host[i] & 2)
        return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "on-device entropy decoding: a DC coefficient leaves the 16 bit coefficient store (damaged stream); the host decoder keeps 32-bit coefficients for it");
      if (walk_status_host[i]) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "speculative decoding settled on something that is not a decode of the image");
    }
  d->phase_device = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb1).count();  // upload + kernel + status
  return evaluate_entropy_status(d, hosts, n, status_host);
}

// ------------------------------------------------------------------------------------------------
// Progressive frames and frames with hidden refinement scans on the device (huffman_prog_kernel)
// ------------------------------------------------------------------------------------------------
// One frame of a file: its decoder, the element type of its planes, where they start in the object's coefficient store.
struct MultiScanFrame {
  HostDecoder *h;
  bool wide;        // int32 coefficients (JPEG XT residual frames with hidden bits)
  int64_t base16;   // offset of the frame's planes in coef_dev, in int16 units
};

// nullptr: every scan of the frame can be decoded one restart interval per lane.
static const char *multiscan_obstacle(const HostDecoder &h, bool xt_part, bool residual_frame)
{
  const mijpeg_info &f = h.info;
  if (h.needs_sequential())
    return "on-device entropy decoding: the stream is damaged; the host decoder walks it with the reference's resynchronisation (entropyparser.cpp:117-201)";
  if (f.xt && !xt_part) return "on-device entropy decoding: not for this JPEG XT stream";
  if (!h.residual_merged()) return "on-device entropy decoding: the legacy codestream has no EOI marker (the host decoder decides what is merged)";
  if (h.verdict_pending()) return "on-device entropy decoding: the file's verdict is the host decoder's (residual codestream header / tables looked up at the first request)";
  if (f.dnl) return "on-device entropy decoding: frames whose height arrives in a DNL marker are decoded on the host";
  // (12-bit frames: the same int16 store as the host decoder's, a coefficient beyond it sends the frame there like everywhere)
  if (f.precision < 8 || f.precision > 12 || (xt_part && !residual_frame && f.precision != 8))
    return "on-device entropy decoding: frames of 8 to 12 bits (JPEG XT: an 8-bit legacy frame)";
  if (h.scans.empty() || h.scans.size() > 4096) return "on-device entropy decoding: no scans, or more than the device path plans for";
  if (!h.every_component_seen()) return "on-device entropy decoding: a component appears in no scan (the host decoder supplies its stand-in)";
  for (int c = 0; c < f.components; c++)
    if (f.hsamp[c] > 4 || f.vsamp[c] > 4) return "on-device entropy decoding: MCUs of more than 4 x 4 blocks of a component are decoded on the host";
  for (size_t si = 0; si < h.scans.size(); si++) {
    const Scan &s = h.scans[si];
    if (s.residual) return "on-device entropy decoding: the residual scan types of part 8 are decoded on the host";
    if (s.ncomp < 1 || (s.se > 0 && s.ss > 0 && s.ncomp != 1)) return "on-device entropy decoding: scan layout";
    if (s.ah > 0 && !s.refinement) return "on-device entropy decoding: scan layout";
    if (s.unstuffed_size >= ((size_t)1 << 28)) return "on-device entropy decoding: entropy coded segment too large for the device decoder's bit addresses";
    const int64_t total_mcus = (int64_t)s.mcus_x * s.mcus_y;
    if (total_mcus < 1 || total_mcus > 0x7fffffff) return "on-device entropy decoding: scan layout";
    if (s.restart_interval > 0) {
      const int64_t nint = (total_mcus + s.restart_interval - 1) / s.restart_interval;
      if ((int64_t)s.interval_begin.size() < nint || (int64_t)s.interval_ubegin.size() < nint)
        return "restart markers missing: the host decoder resynchronises like the reference (entropyparser.cpp:117-201)";
      const std::vector<uint8_t> &rst = h.restart_codes(si);
      if ((int64_t)rst.size() + 1 < nint) return "restart markers missing: the host decoder resynchronises like the reference (entropyparser.cpp:117-201)";
      for (int64_t k = 0; k + 1 < nint; k++)
        if (rst[(size_t)k] != 0xd0 + (k & 7))
          return "restart markers out of sequence: the host decoder resynchronises like the reference (entropyparser.cpp:117-201)";
    } else {
      // One interval: one lane decodes the whole scan.  First passes could be cut into pieces that fall into step with the real
      // decoder; an AC refinement scan cannot -- the bits a block takes depend on which block it is -- so scans
      // without restart markers are left to the host's pipeline of scans unless they are small
      if (s.interval_ubegin.empty()) return "on-device entropy decoding: scan without data";
      if (s.unstuffed_size > ((size_t)24 << 10))
        return "on-device entropy decoding: progressive / refinement scans without restart markers are serial by construction (refinementscan.cpp:584-700): host";
    }
    for (int k = 0; k < s.ncomp; k++) {
      if (s.ss == 0 && s.ah == 0 && !s.dc[k].built) return "on-device entropy decoding: a Huffman table the scan names does not exist";
      if (s.se > 0 && !s.ac[k].built) return "on-device entropy decoding: a Huffman table the scan names does not exist";
    }
  }
  return nullptr;
}

// All scans of the given frames (one file: a progressive picture, or the two frames of a JPEG XT file): upload of the entropy
// coded data without its stuffing, planes cleared, the scans launched level by level (scans that share a component one after
// the other, the rest side by side), range pass.  MIJPEG_ERR_NOT_AVAILABLE: the host decoder's.
static int device_entropy_multiscan(mijpeg_decoder *d, const MultiScanFrame *frames, int nframes, int min_intervals)
{
  struct Item { int frame; size_t scan; int level; int64_t nint; size_t stream_off; size_t table_off; int ntab; int dc_tab[4], ac_tab[4]; int64_t first; };
  std::vector<Item> items;
  const auto tm0 = std::chrono::steady_clock::now();
  static const bool trace_phases = getenv("MIJPEG_TRACE_SUBMIT") != nullptr; // diagnostics: host time of the steps below, on stderr
  auto mark = [&](const char *what) {
    if (trace_phases) fprintf(stderr, "[mijpeg multiscan] %-28s %8.3f ms\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tm0).count() * 1e3);
  };
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t stream_bytes = 0, table_bytes = 0;
  int64_t total_intervals = 0;
  int max_tables = 1;
  std::vector<int> frame_levels((size_t)nframes, 0);
  for (int fi = 0; fi < nframes; fi++) {
    const HostDecoder &h = *frames[fi].h;
    std::vector<int> level(h.scans.size(), 0);
    for (size_t j = 0; j < h.scans.size(); j++) {
      const Scan &b = h.scans[j];
      // (a scan of the AC kind writes whole blocks back: two scans that share a component never run side by side)
      for (size_t i = 0; i < j; i++) {
        const Scan &a = h.scans[i];
        bool common = false;
        for (int ka = 0; ka < a.ncomp; ka++)
          for (int kb = 0; kb < b.ncomp; kb++) common |= a.sc[ka].comp == b.sc[kb].comp;
        if (common) level[j] = std::max(level[j], level[i] + 1);
      }
      Item it;
      memset(&it, 0, sizeof(it));
      it.frame = fi;
      it.scan = j;
      it.level = level[j];
      const int64_t total_mcus = (int64_t)b.mcus_x * b.mcus_y;
      it.nint = b.restart_interval > 0 ? (total_mcus + b.restart_interval - 1) / b.restart_interval : 1;
      it.table_off = table_bytes;
      for (int k = 0; k < b.ncomp; k++) {
        it.dc_tab[k] = it.ac_tab[k] = 0;
        if (b.ss == 0 && b.ah == 0) it.dc_tab[k] = it.ntab++;
        if (b.se > 0) it.ac_tab[k] = it.ntab++;
      }
      table_bytes += (size_t)it.ntab * sizeof(HuffDevTable);
      max_tables = std::max(max_tables, it.ntab);
      it.first = total_intervals;
      total_intervals += it.nint;
      frame_levels[(size_t)fi] = std::max(frame_levels[(size_t)fi], level[j] + 1);
      items.push_back(it);
    }
  }
  // The entropy coded data lies in the upload level by level: what the first launches read goes up first, and the rest is
  // gathered and uploaded while they run (level_end[l]: end of level l's bytes).
  int n_levels = 0;
  for (int fi = 0; fi < nframes; fi++) n_levels = std::max(n_levels, frame_levels[(size_t)fi]);
  std::vector<size_t> level_end((size_t)n_levels, 0);
  for (int lv = 0; lv < n_levels; lv++) {
    for (Item &it : items)
      if (it.level == lv) {
        it.stream_off = stream_bytes;
        stream_bytes += align16(frames[it.frame].h->scans[it.scan].unstuffed_size) + HUFF_STREAM_PAD;
      }
    level_end[(size_t)lv] = stream_bytes;
  }
  if (stream_bytes > 0xfffffff0ull || total_intervals > 0x7fffffff) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "file too large for one device decode");
  // the largest launch decides whether the device is worth the trip ("auto"); every launch -- the scans of one level of one
  // frame -- picks how many lanes of a wave decode by its own number of intervals: fewer lanes = more waves, less divergence
  int lanes_env = 0;
  if (const char *e = getenv("MIJPEG_HUFF_LANES")) { // tuning
    const int l = atoi(e);
    if (l >= 1 && l <= 64 && (l & (l - 1)) == 0) lanes_env = l;
  }
  const int waves = 4;
  int64_t widest = 0, n_groups = 0;
  std::vector<std::vector<int>> level_lanes((size_t)nframes);
  for (int fi = 0; fi < nframes; fi++)
    for (int lv = 0; lv < frame_levels[(size_t)fi]; lv++) {
      int64_t n = 0;
      for (const Item &it : items)
        if (it.frame == fi && it.level == lv) n += it.nint;
      widest = std::max(widest, n);
      int lanes = 64;
      while (lanes > 1 && n / lanes < 768) lanes >>= 1;
      if (lanes_env) lanes = lanes_env;
      level_lanes[(size_t)fi].push_back(lanes);
      for (const Item &it : items)
        if (it.frame == fi && it.level == lv) n_groups += (it.nint + lanes * waves - 1) / (lanes * waves);
    }
  if (min_intervals <= 0) min_intervals = 2048;
  if (widest < min_intervals) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "too few restart intervals to occupy the device");
  // device buffer: [streams][ibegin][iend][tables][scans][groups][status: 8 dwords per frame]
  const size_t off_ib = stream_bytes, off_ie = off_ib + (size_t)total_intervals * 4, off_tab = align16(off_ie + (size_t)total_intervals * 4);
  const size_t off_scan = align16(off_tab + table_bytes), off_grp = align16(off_scan + items.size() * sizeof(ProgScanDev));
  const size_t off_status = align16(off_grp + (size_t)n_groups * sizeof(ProgGroup)), status_bytes = (size_t)nframes * 32, total = off_status + status_bytes;
  int rc = ensure_dev(d, (void **)&d->ent_dev, &d->ent_cap, total);
  if (rc) return rc;
  const size_t host_part = off_status - stream_bytes;
  if (d->ent_host_cap < host_part + status_bytes) {
    if (d->ent_host) (void)hipHostFree(d->ent_host);
    d->ent_host = nullptr;
    d->ent_host_cap = 0;
    HIP_TRY(d, hipHostMalloc((void **)&d->ent_host, host_part + status_bytes, hipHostMallocDefault));
    d->ent_host_cap = host_part + status_bytes;
  }
  rc = ensure_stage(d, stream_bytes);
  if (rc) return rc;
  uint8_t *hp = d->ent_host - stream_bytes; // hp + device offset = staging address
  uint32_t *ib = (uint32_t *)(hp + off_ib), *ie = (uint32_t *)(hp + off_ie);
  ProgScanDev *sd = (ProgScanDev *)(hp + off_scan);
  ProgGroup *groups = (ProgGroup *)(hp + off_grp);
  // groups in launch order: frame, level, scan
  std::vector<std::pair<int64_t, int64_t>> launches; // [first group, groups) of every (frame, level)
  std::vector<int> launch_frame, launch_level;
  int64_t g = 0;
  for (int fi = 0; fi < nframes; fi++)
    for (int lv = 0; lv < frame_levels[(size_t)fi]; lv++) {
      const int64_t g0 = g;
      const int per_group = level_lanes[(size_t)fi][(size_t)lv] * waves;
      for (size_t ii = 0; ii < items.size(); ii++) {
        const Item &it = items[ii];
        if (it.frame != fi || it.level != lv) continue;
        for (int64_t k = 0; k < it.nint; k += per_group) {
          groups[g].scan = (uint32_t)ii;
          groups[g].first_interval = (uint32_t)k;
          g++;
        }
      }
      if (g > g0) { launches.push_back(std::make_pair(g0, g - g0)); launch_frame.push_back(fi); launch_level.push_back(lv); }
    }
  for (size_t ii = 0; ii < items.size(); ii++) {
    const Item &it = items[ii];
    const HostDecoder &h = *frames[it.frame].h;
    const mijpeg_info &f = h.info;
    const Scan &s = h.scans[it.scan];
    ProgScanDev &o = sd[ii];
    memset(&o, 0, sizeof(o));
    o.stream_off = (uint32_t)it.stream_off;
    o.first_interval = (uint32_t)it.first;
    o.n_intervals = (int32_t)it.nint;
    o.total_mcus = s.mcus_x * s.mcus_y;
    o.restart_interval = s.restart_interval > 0 ? s.restart_interval : o.total_mcus;
    o.mcus_x = s.mcus_x;
    o.ncomp = s.ncomp;
    o.ntables = it.ntab;
    o.table_off = (uint32_t)it.table_off;
    o.ss = s.ss; o.se = s.se; o.ah = s.ah; o.al = s.al;
    o.runs_legal = s.progressive_run ? 1 : 0;
    HuffDevTable *tabs = (HuffDevTable *)(hp + off_tab + it.table_off);
    for (int k = 0; k < s.ncomp; k++) {
      const int c = s.sc[k].comp;
      o.comp[k] = c;
      o.hs[k] = s.ncomp > 1 ? f.hsamp[c] : 1;
      o.vs[k] = s.ncomp > 1 ? f.vsamp[c] : 1;
      o.bw[k] = f.blocks_w[c];
      o.coef_off[k] = f.coef_offset[c] / (f.coef_wide ? 2 : 1);
      o.dc_tab[k] = it.dc_tab[k];
      o.ac_tab[k] = it.ac_tab[k];
      if (s.ss == 0 && s.ah == 0) build_dev_table(tabs[it.dc_tab[k]], s.dc[k], 0);
      if (s.se > 0) build_dev_table(tabs[it.ac_tab[k]], s.ac[k], 2);
    }
    memcpy(ib + it.first, s.interval_ubegin.data(), (size_t)it.nint * sizeof(uint32_t)); // (multiscan_obstacle: both lists hold nint entries at least)
    memcpy(ie + it.first, s.interval_uend.data(), (size_t)it.nint * sizeof(uint32_t));
  }
  mark("tables + intervals");
  // The entropy coded data of every scan without its stuffing, gathered by the pool in two goes: what the first launches read
  // (level 0 of every frame), then the rest -- while the copy engine brings up the first part and the first launches run.
  // Uploads on the copy stream, one event per level; the frames' launches wait for their level's event.
  auto gather = [&](int lv0, int lv1) {
    struct Job { size_t item; HostDecoder::UnstuffPiece piece; };
    std::vector<Job> jobs;
    std::vector<HostDecoder::UnstuffPiece> ps;
    for (size_t ii = 0; ii < items.size(); ii++) {
      if (items[ii].level < lv0 || items[ii].level >= lv1) continue;
      ps.clear();
      frames[items[ii].frame].h->unstuff_pieces(items[ii].scan, (size_t)256 << 10, ps);
      for (const auto &pc : ps) jobs.push_back(Job{ii, pc});
    }
    if (jobs.empty()) return;
    const int workers = std::max(1, std::min<int>((int)jobs.size(), std::min(default_threads(), 32)));
    if (trace_phases) fprintf(stderr, "[mijpeg multiscan]   gather of levels %d..%d: %zu pieces on %d workers\n", lv0, lv1 - 1, jobs.size(), workers);
    parallel_for(workers, [&](int w) {
      for (size_t k = (size_t)w; k < jobs.size(); k += (size_t)workers) {
        const Item &it = items[jobs[k].item];
        frames[it.frame].h->unstuff_piece(it.scan, jobs[k].piece, d->stage_host + it.stream_off);
      }
    });
  };
  if (!d->copy_stream) HIP_TRY(d, hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
  while (d->copy_events.size() < (size_t)n_levels) {
    hipEvent_t e = nullptr;
    HIP_TRY(d, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    d->copy_events.push_back(e);
  }
  if (d->ent_free_valid) HIP_TRY(d, hipStreamWaitEvent(d->copy_stream, d->ent_free, 0));
  auto upload_levels = [&](int lv0, int lv1) -> int {
    for (int lv = lv0; lv < lv1; lv++) {
      const size_t b0 = lv ? level_end[(size_t)lv - 1] : 0, b1 = level_end[(size_t)lv];
      if (lv == 0) HIP_TRY(d, hipMemcpyAsync(d->ent_dev + stream_bytes, d->ent_host, host_part, hipMemcpyHostToDevice, d->copy_stream));
      if (b1 > b0) HIP_TRY(d, hipMemcpyAsync(d->ent_dev + b0, d->stage_host + b0, b1 - b0, hipMemcpyHostToDevice, d->copy_stream));
      HIP_TRY(d, hipEventRecord(d->copy_events[(size_t)lv], d->copy_stream));
    }
    return 0;
  };
  ProgArgs a;
  memset(&a, 0, sizeof(a));
  a.data = d->ent_dev;
  a.ibegin = (const uint32_t *)(d->ent_dev + off_ib);
  a.iend = (const uint32_t *)(d->ent_dev + off_ie);
  a.scans = (const ProgScanDev *)(d->ent_dev + off_scan);
  a.waves_per_group = waves;
  a.max_tables = max_tables;
  a.tables = d->ent_dev + off_tab;
  // the two frames of a JPEG XT file share nothing: the second one's launches go to a stream of their own
  hipStream_t second = d->stream;
  if (nframes > 1) {
    if (!d->ms_stream) HIP_TRY(d, hipStreamCreateWithFlags(&d->ms_stream, hipStreamNonBlocking));
    if (!d->ms_ready) HIP_TRY(d, hipEventCreateWithFlags(&d->ms_ready, hipEventDisableTiming));
    if (!d->ms_done) HIP_TRY(d, hipEventCreateWithFlags(&d->ms_done, hipEventDisableTiming));
    second = d->ms_stream;
  }
  auto launch_levels = [&](int lv0, int lv1) -> int {
    for (size_t li = 0; li < launches.size(); li++) {
      const int fi = launch_frame[li], lv = launch_level[li];
      if (lv < lv0 || lv >= lv1) continue;
      hipStream_t st = fi == 0 ? d->stream : second;
      HIP_TRY(d, hipStreamWaitEvent(st, d->copy_events[(size_t)lv], 0));
      a.groups = (const ProgGroup *)(d->ent_dev + off_grp) + launches[li].first;
      a.n_groups = (int32_t)launches[li].second;
      a.lanes = level_lanes[(size_t)fi][(size_t)lv];
      a.wide = frames[fi].wide ? 1 : 0;
      a.coef = (void *)(d->coef_dev + frames[fi].base16);
      a.status = (uint32_t *)(d->ent_dev + off_status) + 8 * fi;
      if (launch_huffman_prog(a, st)) return hip_fail(d, hipGetLastError(), "huffman_prog_kernel launch");
    }
    return 0;
  };
  // where to cut: behind the first level that brings a quarter of the bytes (a progressive frame's DC scan alone is over before
  // anything could hide behind it); no cut when that is the last level
  int cut = n_levels;
  for (int lv = 0; lv + 1 < n_levels; lv++)
    if (level_end[(size_t)lv] * 4 >= stream_bytes) { cut = lv + 1; break; }
  const char *split_env = getenv("MIJPEG_MS_SPLIT"); // A-B: 0 = one gather, then everything enqueued
  if (split_env && atoi(split_env) == 0) cut = n_levels;
  gather(0, cut);
  mark("first levels gathered");
  if ((rc = upload_levels(0, cut))) return rc;
  HIP_TRY(d, hipMemsetAsync(d->ent_dev + off_status, 0, status_bytes, d->stream));
  if (nframes > 1) { // (behind whatever the object's stream still does with the planes, and the cleared status words)
    HIP_TRY(d, hipEventRecord(d->ms_ready, d->stream));
    HIP_TRY(d, hipStreamWaitEvent(second, d->ms_ready, 0));
  }
  // coefficients accumulate over the scans: the planes start out as zeros (coding/blockrow.cpp:77-87)
  for (int fi = 0; fi < nframes; fi++) {
    const mijpeg_info &f = frames[fi].h->info;
    int64_t count = 0;
    for (int c = 0; c < f.components; c++) count += (int64_t)f.blocks_w[c] * f.blocks_h[c] * 64;
    HIP_TRY(d, hipMemsetAsync(d->coef_dev + frames[fi].base16, 0, (size_t)count * (frames[fi].wide ? 4 : 2), fi == 0 ? d->stream : second));
  }
  if ((rc = launch_levels(0, cut))) return rc;
  if (cut < n_levels) {
    gather(cut, n_levels);
    mark("other levels gathered");
    if ((rc = upload_levels(cut, n_levels))) return rc;
    if ((rc = launch_levels(cut, n_levels))) return rc;
  }
  if (second != d->stream) {
    HIP_TRY(d, hipEventRecord(d->ms_done, second));
    HIP_TRY(d, hipStreamWaitEvent(d->stream, d->ms_done, 0));
  }
  for (int fi = 0; fi < nframes; fi++) {
    const mijpeg_info &f = frames[fi].h->info;
    CoefRangeArgs r;
    memset(&r, 0, sizeof(r));
    r.coef = (const void *)(d->coef_dev + frames[fi].base16);
    r.wide = frames[fi].wide ? 1 : 0;
    r.ncomp = f.components;
    for (int c = 0; c < f.components; c++) {
      r.coef_off[c] = f.coef_offset[c] / (f.coef_wide ? 2 : 1);
      r.nblocks[c] = (int64_t)f.blocks_w[c] * f.blocks_h[c];
      memcpy(r.q[c], f.quant[f.quant_index[c]], sizeof(r.q[c]));
    }
    r.status = (uint32_t *)(d->ent_dev + off_status) + 8 * fi;
    if (launch_coef_range(r, d->stream)) return hip_fail(d, hipGetLastError(), "coef_range_kernel launch");
  }
  uint32_t *status_host = (uint32_t *)(d->ent_host + host_part);
  HIP_TRY(d, hipMemcpyAsync(status_host, d->ent_dev + off_status, status_bytes, hipMemcpyDeviceToHost, d->stream));
  if (!d->ent_free) HIP_TRY(d, hipEventCreateWithFlags(&d->ent_free, hipEventDisableTiming));
  HIP_TRY(d, hipEventRecord(d->ent_free, d->stream));
  d->ent_free_valid = true;
  mark("launches enqueued");
  HIP_TRY(d, hipStreamSynchronize(d->stream));
  mark("device done");
  std::vector<HostDecoder *> hosts((size_t)nframes);
  for (int fi = 0; fi < nframes; fi++) hosts[(size_t)fi] = frames[fi].h;
  return evaluate_entropy_status(d, hosts.data(), nframes, status_host);
}

int mijpeg_decode_coefficients_device(mijpeg_decoder *d, int min_intervals)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (!d->data) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no input stream has been set");
  if (d->device < 0) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "decoder was created without a device");
  HIP_TRY(d, hipSetDevice(d->device));
  if (const int prc = settle_pending(d)) return prc;
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  d->parse_fresh = false;
  int rc = d->host.parse(d->data, d->size, false);
  if (rc) return set_error(d, rc, d->host.error.message);
  d->parsed = true;
  d->batch_frames = 0;
  const auto t_parsed = clk::now();
  HostDecoder *h = &d->host, *res = d->host.residual();
  // (a stream that does not qualify: the parse is as good as the one mijpeg_decode_coefficients would make next)
  d->parse_fresh = true;
  // progressive frames and frames with hidden refinement scans: every scan one restart interval per lane (huffman_prog_kernel)
  auto many_scans = [](const HostDecoder &x) { return x.info.progressive != 0 || x.has_hidden_scans() || x.scans.size() != 1; };
  static const bool no_multiscan = getenv("MIJPEG_NO_DEVICE_MULTISCAN") != nullptr; // A-B comparisons
  // (... and 12-bit frames, whose single scan the sequential kernel's path declines: round 6)
  const bool multiscan = !no_multiscan && (many_scans(d->host) || (res && many_scans(*res)) || (!res && d->host.info.precision != 8));
  if (multiscan) {
    if (const char *why = multiscan_obstacle(d->host, res != nullptr, false)) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, why);
    if (res)
      if (const char *why = multiscan_obstacle(*res, true, true)) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, why);
  } else {
    if (const char *why = device_entropy_obstacle(d->host, d->size, res != nullptr)) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, why);
    if (res) {
      if (const char *why = device_entropy_obstacle(*res, res->stream_size(), true)) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, why);
      if (d->host.xt.residual_wide) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "32-bit residual coefficients are decoded on the host");
    }
  }
  d->parse_fresh = false;
  rc = ensure_coef_store(d, (size_t)d->host.info.coef_count, false);
  if (rc) return rc;
  d->img_valid = d->model_valid = false;
  d->uploaded = false;
  d->decoded = false;
  // JPEG XT: the planes of the residual frame follow those of the legacy frame in the same store
  int64_t own_count = 0;
  for (int c = 0; c < d->host.info.components; c++) own_count += (int64_t)d->host.info.blocks_w[c] * d->host.info.blocks_h[c] * 64;
  static const bool trace_read = getenv("MIJPEG_TRACE_SUBMIT") != nullptr; // diagnostics
  if (trace_read)
    fprintf(stderr, "[mijpeg device read] parse %.3f ms, checks + coefficient store %.3f ms\n", std::chrono::duration<double>(t_parsed - t0).count() * 1e3,
            std::chrono::duration<double>(clk::now() - t_parsed).count() * 1e3);
  if (multiscan) {
    MultiScanFrame fr[2] = {{h, false, 0}, {res, res && d->host.xt.residual_wide != 0, own_count}};
    rc = device_entropy_multiscan(d, fr, res ? 2 : 1, min_intervals);
    if (!rc && res) {
      for (int c = 0; c < MIJPEG_MAX_COMPONENTS; c++) d->host.xt.residual.range_max[c] = res->info.range_max[c];
      d->host.info.fast_arith = 0; // as HostDecoder::decode has it: the fast flavours are chosen per kernel for XT
    }
  } else if (!res) {
    rc = device_entropy_batch(d, &h, &d->data, &d->size, 1, min_intervals, d->coef_dev, own_count, false);
  } else {
    // JPEG XT: the two codestreams are independent, so the residual one is decoded at the same time by a helper object
    // (own stream, own buffers) on a thread of its own, straight into the planes behind the legacy frame's
    if (!d->xt_helper && mijpeg_create(&d->xt_helper, d->device) != MIJPEG_OK) return set_error(d, MIJPEG_ERR_OUT_OF_MEMORY, "no helper decoder for the residual codestream");
    const uint8_t *rdata = res->stream_base();
    const size_t rsize = res->stream_size();
    int rc2 = MIJPEG_OK;
    std::thread helper([&]() {
      try {
        if (hipSetDevice(d->device) != hipSuccess) { rc2 = MIJPEG_ERR_DEVICE; return; }
        rc2 = device_entropy_batch(d->xt_helper, &res, &rdata, &rsize, 1, min_intervals, d->coef_dev + own_count, res->info.coef_count, true);
      } catch (...) { // (nothing may leave a thread's function)
        rc2 = boundary_catch(d->xt_helper, "residual codestream, device entropy decoding");
      }
    });
    struct Joiner { // (an exception on this thread must not meet a joinable thread object)
      std::thread &t;
      ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{helper};
    rc = device_entropy_batch(d, &h, &d->data, &d->size, 1, min_intervals, d->coef_dev, own_count, true);
    helper.join();
    if (!rc && rc2) {
      const char *m = nullptr;
      mijpeg_last_error(d->xt_helper, &m);
      rc = set_error(d, rc2, m ? m : "residual codestream: device entropy decoding failed");
    }
    if (!rc) {
      for (int c = 0; c < MIJPEG_MAX_COMPONENTS; c++) d->host.xt.residual.range_max[c] = res->info.range_max[c];
      d->host.info.fast_arith = 0; // as HostDecoder::decode has it: the fast flavours are chosen per kernel for XT
    }
  }
  d->timing[0] = std::chrono::duration<double>(clk::now() - t0).count();
  d->timing[1] = std::chrono::duration<double>(t_parsed - t0).count(); // header parse + restart marker search
  d->timing[2] = d->timing[3] = 0;
  if (trace_read) fprintf(stderr, "[mijpeg device read] whole call %.3f ms\n", d->timing[0] * 1e3);
  if (rc) return rc;
  d->decoded = true;
  d->uploaded = true;
  d->host_planes_stale = true;
  return decode_alpha_channel(d, 0);
} catch (...) { return boundary_catch(d, "mijpeg_decode_coefficients_device"); }

// ------------------------------------------------------------------------------------------------
// batches: n streams of one geometry -> n coefficient stores -> n frames, two kernel launches in all
// ------------------------------------------------------------------------------------------------
// Aggregation over the images of a decoded batch: what one reconstruction launch for all of them needs to know.
static int finish_batch(mijpeg_decoder *d);

// What the last finished batch of shared tables reported, for the speculative launch of the next one (MIJPEG_FLAG_SPECULATIVE):
// frame geometry, tables, and the range check that selected its kernel.  Process-wide: the decoder objects of a pipeline work
// on chunks of the same material.
namespace {
struct SpecHint {
  std::mutex m;
  bool valid = false;
  mijpeg_info info{};
};
SpecHint *spec_hint()
{
  static SpecHint *h = new SpecHint;
  return h;
}
// the gates of the kernel selection (use_* above): an assumed range just below the next gate selects the kernel the hint's
// batch ran on and holds for every batch that stays below that gate
int32_t next_gate_below(int32_t range)
{
  static const int32_t gates[] = {1477, 2047, 7600, 8190, 16384, 45056, 49152, 65536}; // (1477: use_dot2_pass's range_max <= 1476)
  for (int32_t g : gates)
    if (range < g) return g - 1;
  return -1;
}
bool same_shape_and_tables(const mijpeg_info &a, const mijpeg_info &b)
{
  if (a.width != b.width || a.height != b.height || a.components != b.components || a.precision != b.precision || a.ycbcr != b.ycbcr || a.xt != b.xt ||
      a.dnl != b.dnl || a.coef_count != b.coef_count)
    return false;
  for (int c = 0; c < a.components; c++) {
    if (a.hsamp[c] != b.hsamp[c] || a.vsamp[c] != b.vsamp[c]) return false;
    if (memcmp(a.quant[a.quant_index[c]], b.quant[b.quant_index[c]], sizeof(a.quant[0]))) return false;
  }
  return true;
}
} // namespace
static int settle_speculation(mijpeg_decoder *d);

static int submit_batch(mijpeg_decoder *d, const uint8_t *const *streams, const size_t *sizes, int n, int min_intervals, bool defer)
{
  if (!d || !streams || !sizes || n < 1) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "decoder was created without a device");
  HIP_TRY(d, hipSetDevice(d->device));
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  if (d->spec_active) { // a speculative reconstruction nobody validated (mijpeg_finish_batch_device): its verdict comes first
    const int src = settle_speculation(d);
    if (src) return src;
  }
  if (const int prc = settle_pending(d)) return prc; // a submitted batch nobody waited for: its staging buffers are about to be reused
  d->batch_frames = 0;
  if (d->batch_hosts.size() < (size_t)n) d->batch_hosts.resize((size_t)n); // never shrinks: a pipeline's chunks differ in size, and
  for (auto &h : d->batch_hosts)                                         // a parser that is thrown away takes its grown vectors along
    if (!h) h.reset(new HostDecoder());
  // headers and restart markers of all streams, one stream per worker -- which writes the device's copy of the entropy
  // coded data (no byte stuffing, no markers) into the stream's slot of the pinned gathering area while it is at it
  std::vector<int> rcs((size_t)n, 0);
  {
    std::vector<size_t> slot;
    const size_t total = stream_slots(sizes, n, slot);
    const int src = ensure_stage(d, total);
    if (src) return src;
    parallel_for(std::min(n, default_threads()), [&](int w) {
      for (int i = w; i < n; i += std::min(n, default_threads())) {
        // (a worker walks ~3 GB/s this way: good for the many small streams of a batch; a large stream is searched in
        // parallel chunks and gathered in parallel pieces instead -- device_entropy_batch sees which it was)
        if (sizes[i] <= ((size_t)2 << 20) || n >= default_threads()) d->batch_hosts[(size_t)i]->set_unstuff_sink(d->stage_host + slot[(size_t)i], sizes[i]);
        rcs[(size_t)i] = d->batch_hosts[(size_t)i]->parse(streams[i], sizes[i], false);
      }
    });
  }
  for (int i = 0; i < n; i++)
    if (rcs[(size_t)i]) return set_error(d, rcs[(size_t)i], d->batch_hosts[(size_t)i]->error.message);
  const auto t_parsed = clk::now();
  std::vector<HostDecoder *> hosts((size_t)n);
  for (int i = 0; i < n; i++) hosts[(size_t)i] = d->batch_hosts[(size_t)i].get();
  for (int i = 0; i < n; i++)
    if (const char *why = device_entropy_obstacle(*hosts[(size_t)i], sizes[i])) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, why);
  const mijpeg_info &f0 = hosts[0]->info;
  // one reconstruction launch serves the batch; images with tables of their own (motion JPEG under rate control) make it
  // read per-frame tables from device memory instead of the kernel arguments
  bool own_tables = false;
  for (int i = 1; i < n && !own_tables; i++) {
    const mijpeg_info &f = hosts[(size_t)i]->info;
    for (int c = 0; c < f.components; c++)
      if (memcmp(f.quant[f.quant_index[c]], f0.quant[f0.quant_index[c]], sizeof(f.quant[0]))) own_tables = true;
  }
  int rc = ensure_coef_store(d, (size_t)f0.coef_count * (size_t)n, false);
  if (rc) return rc;
  d->img_valid = d->model_valid = false;
  d->uploaded = false;
  d->decoded = false;
  d->pend_n = 0;
  rc = device_entropy_batch(d, hosts.data(), streams, sizes, n, min_intervals, d->coef_dev, f0.coef_count, false, defer);
  d->timing[0] = std::chrono::duration<double>(clk::now() - t0).count();
  d->timing[1] = std::chrono::duration<double>(t_parsed - t0).count();
  d->timing[2] = d->phase_prepare;
  d->timing[3] = d->phase_device;
  if (rc) { // copies may have been enqueued before the failure: nothing of this batch stays in flight
    d->pend_n = 1;
    (void)settle_pending(d);
    return rc;
  }
  d->batch_own_tables = own_tables;
  d->batch_frames = -n; // decoded (or on its way) but not aggregated yet
  if (d->pend_n) return MIJPEG_OK; // deferred: finish_batch() waits
  return finish_batch(d);
}

static int finish_batch(mijpeg_decoder *d)
{
  const int n = d->batch_frames < 0 ? -d->batch_frames : 0;
  if (n == 0) return d->batch_frames > 0 ? MIJPEG_OK : set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no batch has been submitted");
  d->batch_frames = 0;
  std::vector<HostDecoder *> hosts((size_t)n);
  for (int i = 0; i < n; i++) hosts[(size_t)i] = d->batch_hosts[(size_t)i].get();
  if (d->pend_n) { // wait for the upload and the Huffman kernel of the submitted batch, then look at what it reported
    const int pn = d->pend_n;
    d->pend_n = 0;
    // The pipeline's one wait on the device.  hipStreamSynchronize blocks on an interrupt after a short spin, and how long the wake-up
    // takes is the host's business (idle states of the core that takes the interrupt): on some boxes 0.3 ms per wait for seconds
    // on end -- sixteen chunks of a batch, five milliseconds (profiles/r05/batch4k_stall.txt).  A submitted batch is a
    // millisecond from done when somebody asks for it: poll the stream for that long, block only beyond.
    {
      static const bool no_spin = getenv("MIJPEG_NO_SPIN_WAIT") != nullptr; // A-B measurements
      const auto t_spin = std::chrono::steady_clock::now();
      hipError_t q = hipSuccess;
      while (!no_spin && (q = hipStreamQuery(d->stream)) == hipErrorNotReady) {
        if (std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(4)) break;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      // hipErrorNotReady is not an error: drop it -- and only it; anything else the query saw (a failed launch of the work it waits
      // for) is the batch's verdict
      if (q == hipErrorNotReady) (void)hipGetLastError();
      else if (q != hipSuccess) HIP_TRY(d, q);
    }
    HIP_TRY(d, hipStreamSynchronize(d->stream));
    d->phase_device += std::chrono::duration<double>(std::chrono::steady_clock::now() - d->pend_t0).count();
    if (d->pend_walk_round > 0) { // streams without restart markers: did the walk settle within the rounds it was given?
      const int rounds = d->pend_walk_round;
      d->pend_walk_round = 0;
      if (d->pend_walk_flags[rounds]) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "speculative decoding did not settle in the rounds a submitted batch gets: decode it with mijpeg_decode_batch_device");
      d->walk_rounds = 1;
      for (int r = 1; r <= rounds; r++)
        if (d->pend_walk_flags[r]) d->walk_rounds = r + 1;
      for (int i = 0; i < pn && d->pend_walk_status; i++) {
        if (d->pend_walk_status[i] & 2)
          return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "on-device entropy decoding: a DC coefficient leaves the 16 bit coefficient store (damaged stream); the host decoder keeps 32-bit coefficients for it");
        if (d->pend_walk_status[i]) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "speculative decoding settled on something that is not a decode of the image");
      }
    }
    const int rc = evaluate_entropy_status(d, hosts.data(), pn, d->pend_status);
    if (rc) return rc;
  }
  const mijpeg_info &f0 = hosts[0]->info;
  const bool own_tables = d->batch_own_tables;
  int rc = MIJPEG_OK;
  d->batch_info = f0;
  d->batch_own_tables = own_tables;
  if (own_tables) {
    // the batch's info then carries, per COMPONENT, the largest delta any image has at each position: what the range
    // gates of the kernel selection look at
    d->batch_quant_host.assign((size_t)n * 4 * 64, 1);
    mijpeg_info &bi = d->batch_info;
    for (int c = 0; c < f0.components; c++) {
      bi.quant_index[c] = (uint8_t)c;
      for (int k = 0; k < 64; k++) bi.quant[c][k] = 0;
    }
    for (int i = 0; i < n; i++) {
      const mijpeg_info &f = hosts[(size_t)i]->info;
      for (int c = 0; c < f.components; c++)
        for (int k = 0; k < 64; k++) {
          const uint16_t q = f.quant[f.quant_index[c]][k];
          d->batch_quant_host[((size_t)i * 4 + c) * 64 + k] = q;
          bi.quant[c][k] = std::max(bi.quant[c][k], q);
        }
    }
    const size_t bytes = d->batch_quant_host.size() * sizeof(uint16_t);
    rc = ensure_dev(d, (void **)&d->batch_quant_dev, &d->batch_quant_cap, bytes);
    if (rc) return rc;
    HIP_TRY(d, hipMemcpyAsync(d->batch_quant_dev, d->batch_quant_host.data(), bytes, hipMemcpyHostToDevice, d->stream));
  }
  d->batch_info.fast_arith = 1;
  for (int i = 0; i < n; i++) { // the batch is as fast as its most demanding image
    const mijpeg_info &f = hosts[(size_t)i]->info;
    if (!f.fast_arith) d->batch_info.fast_arith = 0;
    for (int c = 0; c < f.components; c++) d->batch_info.range_max[c] = std::max(d->batch_info.range_max[c], f.range_max[c]);
  }
  d->batch_frames = n;
  if (!own_tables && !d->batch_info.xt) { // the next batch of this shape may launch its reconstruction on this range check
    SpecHint &h = *spec_hint();
    std::lock_guard<std::mutex> lock(h.m);
    h.info = d->batch_info;
    h.valid = true;
  }
  return MIJPEG_OK;
}

// A speculative launch is validated: the batch is finished the ordinary way (wait, errors, range check), and where the range
// check is not the one the launch assumed the reconstruction runs again with the kernel the real one selects.
static int settle_speculation(mijpeg_decoder *d)
{
  if (!d->spec_active) return MIJPEG_OK;
  d->spec_active = false;
  d->spec_redone = false;
  if (d->batch_frames == 0) return MIJPEG_OK; // (the batch was abandoned: another stream was set on the object)
  if (d->batch_frames < 0) {
    const int rc = finish_batch(d);
    if (rc) return rc; // (the stream is damaged, the walk had not settled ...: nothing of the speculative pixels counts)
  }
  bool holds = d->batch_info.fast_arith != 0;
  for (int c = 0; c < d->batch_info.components; c++) holds = holds && d->batch_info.range_max[c] <= d->spec_assumed[c];
  if (holds) return MIJPEG_OK;
  d->spec_redone = true;
  d->spec_redone_count++;
  return mijpeg_reconstruct_batch_device(d, d->spec_dst, d->spec_frame_stride, d->spec_row_stride, d->spec_flags & ~MIJPEG_FLAG_SPECULATIVE, 1);
}

int mijpeg_decode_batch_device(mijpeg_decoder *d, const uint8_t *const *streams, const size_t *sizes, int n, int min_intervals)
try {
  return submit_batch(d, streams, sizes, n, min_intervals, false);
} catch (...) { return boundary_catch(d, "mijpeg_decode_batch_device"); }

int mijpeg_prepare_batch_host(mijpeg_decoder *d, const uint8_t *const *streams, const size_t *sizes, int n)
try {
  if (!d || !streams || !sizes || n < 1) return MIJPEG_ERR_INVALID_PARAMETER;
  if (const int prc = settle_pending(d)) return prc;
  d->batch_frames = 0;
  if (d->batch_hosts.size() < (size_t)n) d->batch_hosts.resize((size_t)n); // never shrinks: a pipeline's chunks differ in size, and
  for (auto &h : d->batch_hosts)                                         // a parser that is thrown away takes its grown vectors along
    if (!h) h.reset(new HostDecoder());
  std::vector<size_t> slot;
  const size_t total = stream_slots(sizes, n, slot);
  uint8_t *stage;
  if (d->device >= 0) {
    HIP_TRY(d, hipSetDevice(d->device));
    if (const int src = ensure_stage(d, total)) return src;
    stage = d->stage_host;
  } else {
    if (d->host_stage.size() < total) d->host_stage.resize(total);
    stage = d->host_stage.data();
  }
  std::vector<int> rcs((size_t)n, 0);
  const int workers = std::min(n, default_threads());
  parallel_for(workers, [&](int w) {
    for (int i = w; i < n; i += workers) {
      d->batch_hosts[(size_t)i]->set_unstuff_sink(stage + slot[(size_t)i], sizes[i]);
      rcs[(size_t)i] = d->batch_hosts[(size_t)i]->parse(streams[i], sizes[i], false);
    }
  });
  for (int i = 0; i < n; i++)
    if (rcs[(size_t)i]) return set_error(d, rcs[(size_t)i], d->batch_hosts[(size_t)i]->error.message);
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_prepare_batch_host"); }

int mijpeg_submit_batch_device(mijpeg_decoder *d, const uint8_t *const *streams, const size_t *sizes, int n, int min_intervals)
try {
  return submit_batch(d, streams, sizes, n, min_intervals, true);
} catch (...) { return boundary_catch(d, "mijpeg_submit_batch_device"); }

int mijpeg_synchronize(mijpeg_decoder *d)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0 || !d->stream) return MIJPEG_OK;
  HIP_TRY(d, hipSetDevice(d->device));
  HIP_TRY(d, hipStreamSynchronize(d->stream));
  if (d->spec_active) return settle_speculation(d); // (pixels of a speculative launch count once it is validated)
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_synchronize"); }

int mijpeg_stream_wait(mijpeg_decoder *d, void *client_stream)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0 || !d->stream) return MIJPEG_OK;
  HIP_TRY(d, hipSetDevice(d->device));
  if (!d->chain_ev) HIP_TRY(d, hipEventCreateWithFlags(&d->chain_ev, hipEventDisableTiming));
  HIP_TRY(d, hipEventRecord(d->chain_ev, d->stream));
  HIP_TRY(d, hipStreamWaitEvent((hipStream_t)client_stream, d->chain_ev, 0));
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_stream_wait"); }

int mijpeg_finish_batch_device(mijpeg_decoder *d)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device >= 0) HIP_TRY(d, hipSetDevice(d->device));
  if (d->spec_active) return settle_speculation(d);
  return finish_batch(d);
} catch (...) { return boundary_catch(d, "mijpeg_finish_batch_device"); }

int mijpeg_batch_speculation(mijpeg_decoder *d, int64_t *launched, int64_t *redone)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (launched) *launched = d->spec_launched;
  if (redone) *redone = d->spec_redone_count;
  return d->spec_redone ? 1 : 0;
} catch (...) { return boundary_catch(d, "mijpeg_batch_speculation"); }

int mijpeg_reconstruct_batch_device(mijpeg_decoder *d, void *dst_device, int64_t frame_stride, int64_t row_stride, uint32_t flags, int sync)
try {
  if (!d || !dst_device) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device >= 0) HIP_TRY(d, hipSetDevice(d->device));
  bool speculate = false;
  mijpeg_info assumed;
  if (d->batch_frames < 0 && (flags & MIJPEG_FLAG_SPECULATIVE) && !sync && d->pend_n && !d->pend_walk_round && !d->batch_own_tables && !d->spec_active) {
    // A submitted batch whose Huffman kernel may still be running: launch the reconstruction behind it on the range check the
    // last batch of this shape and these tables reported (rounded up to the selection's next gate) instead of waiting for this
    // one's.  The pipeline's host thread never blocks on the device; mijpeg_finish_batch_device validates.
    const mijpeg_info &f0 = d->batch_hosts[0]->info;
    SpecHint &h = *spec_hint();
    std::lock_guard<std::mutex> lock(h.m);
    if (h.valid && h.info.fast_arith && same_shape_and_tables(h.info, f0)) {
      assumed = f0;
      assumed.fast_arith = 1;
      speculate = true;
      for (int c = 0; c < f0.components; c++) {
        assumed.range_max[c] = next_gate_below(h.info.range_max[c]);
        if (assumed.range_max[c] < 0) speculate = false;
      }
    }
  }
  if (d->batch_frames < 0 && !speculate) { // submitted with mijpeg_submit_batch_device: wait for it now
    const int rc = d->spec_active ? settle_speculation(d) : finish_batch(d);
    if (rc) return rc;
  }
  if (d->batch_frames < 1 && !speculate) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no decoded batch: call mijpeg_decode_batch_device first");
  mijpeg_batch b;
  memset(&b, 0, sizeof(b));
  b.info = speculate ? assumed : d->batch_info;
  b.coef_dev = d->coef_dev;
  b.coef_frame_stride = b.info.coef_count;
  b.out_dev = (uint8_t *)dst_device;
  b.out_row_stride = row_stride;
  b.out_frame_stride = frame_stride;
  b.frames = speculate ? -d->batch_frames : d->batch_frames;
  b.quant_dev = d->batch_own_tables ? d->batch_quant_dev : nullptr;
  b.flags = flags & ~(MIJPEG_FLAG_DEVICE_OUTPUT | MIJPEG_FLAG_NO_UPSAMPLING | MIJPEG_FLAG_SPECULATIVE);
  const size_t ws = mijpeg_workspace_bytes(&b);
  if (ws) {
    const int rc = ensure_dev(d, (void **)&d->ws_dev, &d->ws_cap, ws);
    if (rc) return rc;
    b.workspace = d->ws_dev;
    b.workspace_bytes = d->ws_cap;
  }
  const int rc = mijpeg_launch_reconstruct(&b, d->stream);
  if (rc) return set_error(d, rc, rc == MIJPEG_ERR_DEVICE ? std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError())
                                                           : std::string("reconstruction not available for this batch"));
  if (speculate) {
    d->spec_active = true;
    d->spec_redone = false;
    d->spec_dst = dst_device;
    d->spec_frame_stride = frame_stride;
    d->spec_row_stride = row_stride;
    d->spec_flags = flags;
    for (int c = 0; c < MIJPEG_MAX_COMPONENTS; c++) d->spec_assumed[c] = assumed.range_max[c];
    d->spec_launched++;
    return MIJPEG_OK;
  }
  if (sync) HIP_TRY(d, hipStreamSynchronize(d->stream));
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_reconstruct_batch_device"); }

const int16_t *mijpeg_device_coefficients(mijpeg_decoder *d) { return (d && d->uploaded) ? d->coef_dev : nullptr; }

int mijpeg_last_error(mijpeg_decoder *d, const char **message)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (message) *message = d->err_code ? d->err_msg.c_str() : nullptr;
  return d->err_code;
} catch (...) { return boundary_catch(d, "mijpeg_last_error"); }

mijpeg_decoder *mijpeg_alpha_channel(mijpeg_decoder *d)
try {
  if (!d) return nullptr;
  if (d->decoded && !d->alpha_ready && d->alpha_refusal) { // (what the reference reports at the first request for alpha pixels)
    set_error(d, d->alpha_refusal, d->alpha_refusal_msg);
    return nullptr;
  }
  if (!d->decoded || !d->alpha_ready) {
    set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "the decoded stream has no alpha channel");
    return nullptr;
  }
  return d->alpha;
} catch (...) { (void)boundary_catch(d, "mijpeg_alpha_channel"); return nullptr; }

int mijpeg_has_alpha(mijpeg_decoder *d)
try {
  return d && d->decoded && d->alpha_ready ? 1 : 0; // (a query: leaves the object's last error alone)
} catch (...) { return boundary_catch(d, "mijpeg_has_alpha"); }

int mijpeg_alpha_info(mijpeg_decoder *d, int32_t *mode, int32_t matte[3])
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (!d->decoded || !d->alpha_ready) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "the decoded stream has no alpha channel");
  if (mode) *mode = d->host.alpha_mode();
  for (int k = 0; k < 3 && matte; k++) matte[k] = (int32_t)d->host.alpha_matte()[k];
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_alpha_info"); }

int mijpeg_last_warning(mijpeg_decoder *d, const char **message)
try {
  if (message) *message = nullptr;
  if (!d || !d->data) return 0;
  return d->host.last_warning(message);
} catch (...) { return boundary_catch(d, "mijpeg_last_warning"); }

int mijpeg_last_timing(mijpeg_decoder *d, double out_seconds[4])
try {
  if (!d || !out_seconds) return MIJPEG_ERR_INVALID_PARAMETER;
  for (int i = 0; i < 4; i++) out_seconds[i] = d->timing[i];
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_last_timing"); }

// ------------------------------------------------------------------------------------------------
// stateless batch launch
// ------------------------------------------------------------------------------------------------
static bool is_420(const mijpeg_info &f)
{
  return f.components == 3 && f.hsamp[0] == 2 && f.vsamp[0] == 2 && f.hsamp[1] == 1 && f.vsamp[1] == 1 &&
         f.hsamp[2] == 1 && f.vsamp[2] == 1;
}

static bool is_444(const mijpeg_info &f)
{
  return f.components == 3 && f.hsamp[0] == 1 && f.vsamp[0] == 1 && f.hsamp[1] == 1 && f.vsamp[1] == 1 && f.hsamp[2] == 1 &&
         f.vsamp[2] == 1;
}

static bool is_422(const mijpeg_info &f)
{
  return f.components == 3 && f.hsamp[0] == 2 && f.vsamp[0] == 1 && f.hsamp[1] == 1 && f.vsamp[1] == 1 && f.hsamp[2] == 1 &&
         f.vsamp[2] == 1;
}

static bool is_411(const mijpeg_info &f)
{
  return f.components == 3 && f.hsamp[0] == 4 && f.vsamp[0] == 1 && f.hsamp[1] == 1 && f.vsamp[1] == 1 && f.hsamp[2] == 1 &&
         f.vsamp[2] == 1;
}

static bool is_440(const mijpeg_info &f)
{
  return f.components == 3 && f.hsamp[0] == 1 && f.vsamp[0] == 2 && f.hsamp[1] == 1 && f.vsamp[1] == 1 && f.hsamp[2] == 1 &&
         f.vsamp[2] == 1;
}

// The fused kernels address inside a frame with 32-bit byte offsets (planes and pixels; frames are 64 bits apart): frames
// beyond that -- a 65535 x 65535 picture has 8.6 GB of luma coefficients and 12.9 GB of pixels -- take the generic kernels,
// whose addressing is 64 bits wide throughout.
// DNL frames (mijpeg_info::dnl): the vertical filter of a subsampled component reads the line below the picture's last one,
// and when the picture ends on a block row boundary that line belongs to the block row the first scan creates behind the
// picture -- unless it met the marker before it got there.  Then the row does not exist, the reference reads NULL and
// transforms it to samples of value 0 (control/blockbitmaprequester.cpp:1097-1108, dct/idct.cpp:336-338): no coefficients
// give that, the unfused kernels write the zeros themselves (GenericArgs::zero_from).
static bool dnl_row_missing(const mijpeg_info &f)
{
  if (!f.dnl) return false;
  for (int c = 0; c < f.components; c++) {
    const int ch = (f.height + f.suby[c] - 1) / f.suby[c];
    if (f.suby[c] > 1 && (ch & 7) == 0 && f.rows[c] <= (ch >> 3)) return true;
  }
  return false;
}

static bool fits32(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  const uint64_t lim = 0xffffffffull;
  if (f.coef_wide) return false; // int32 coefficients (damaged stream): the unfused kernels' business
  if (dnl_row_missing(f)) return false; // (the fused kernels have no way to say "this block row is NULL")
  for (int c = 0; c < f.components; c++)
    if ((uint64_t)f.blocks_w[c] * (uint64_t)f.blocks_h[c] * 128u > lim) return false;
  if (f.xt && b->xt)
    for (int c = 0; c < b->xt->residual.components; c++)
      if ((uint64_t)b->xt->residual.blocks_w[c] * (uint64_t)b->xt->residual.blocks_h[c] * 128u > lim) return false;
  if (b->out_row_stride < 0) return false; // bottom-up bitmaps: the offsets are unsigned
  // (a batch description without strides -- mijpeg_kernel_name, mijpeg_workspace_bytes asked ahead of time -- is taken to
  // have tightly packed lines)
  const uint64_t line = (uint64_t)f.width * (uint64_t)f.components * (f.xt ? (uint64_t)(f.sample_bytes > 1 ? 2 : 1) : f.precision > 8 ? 2u : 1u);
  const uint64_t rs = b->out_row_stride ? (uint64_t)b->out_row_stride : line;
  return (uint64_t)f.height * rs + line <= lim;
}

static bool fast_ok(const mijpeg_batch *b)
{
  // fast arithmetic: range check passed (host decoder) and every delta << 4 fits a signed 16-bit operand
  const mijpeg_info &f = b->info;
  if (!f.fast_arith || (b->flags & MIJPEG_FLAG_FORCE_SAFE)) return false;
  for (int c = 0; c < f.components; c++)
    for (int i = 0; i < 64; i++)
      if (f.quant[f.quant_index[c]][i] > 2047) return false;
  return true;
}

// fused 4:4:4 keeps the chroma samples as packed int16: needs |sample| <= 4 * range_max < 32768
static bool use_fused444(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  return is_444(f) && f.ycbcr && !f.xt && f.precision == 8 && !(b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_NO_COLOR_TRANSFORM)) && fast_ok(b) &&
         f.range_max[1] < 8190 && f.range_max[2] < 8190 && fits32(b);
}

static bool use_fused420(const mijpeg_batch *b)
{
  return is_420(b->info) && b->info.ycbcr && !b->info.xt && b->info.precision == 8 &&
         !(b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_NO_COLOR_TRANSFORM)) && fits32(b);
}

// 12-bit 4:2:0 frames (SOF1, P = 12) inside the ranges the 12-bit flavour of the fused kernel is exact for: every delta << 4 a
// signed 16-bit operand; sum |c| q < 49152 bounds every butterfly intermediate by 1573 * 16 * 49152 < 2^31 (first pass;
// the second pass sees at most 22.2 * range_max per column) and every multiplicand by 2^23; chroma sum |c| q < 45056 bounds
// the chroma samples (times 16) by 4.02 * 45056 + 2 < 181 200 (|basis| <= 1/4 per coefficient, the 9-bit constants and the
// roundings add < 0.5 %), whose products with the colour constants (11485; 2819 + 5850; 14516 taken as 4 * 3629) fit 32 bits
static bool use_fused420_12(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F420_12") != nullptr; // A-B comparisons
  if (off || !is_420(f) || !f.ycbcr || f.xt || f.precision != 12 ||
      (b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_NO_COLOR_TRANSFORM | MIJPEG_FLAG_FORCE_SAFE)) || !fits32(b))
    return false;
  if (f.range_max[0] <= 0 || f.range_max[0] >= 49152 || f.range_max[1] >= 45056 || f.range_max[2] >= 45056) return false;
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < 64; i++)
      if (f.quant[f.quant_index[c]][i] > 2047) return false;
  return true;
}

// The 12-bit kernels' colour stage in one 32-bit sum per channel (colour12<true>, kernels.hip): the luma sample times 16 is at most
// 4.02 * range_max[0] + 2 in magnitude, a chroma sample behind the upsampling filters 4.02 * range_max[c] + 4 (the bounds of
// use_fused420_12; the filters are convex combinations plus a rounding), so (|y'| + 32776) * 8192 + 14516 |c| -- 14516 is the largest
// weight a channel puts on chroma, 2819 + 5850 the green one's -- stays below 2^31 where this holds.  Monotone in every range: a
// speculative launch that assumed larger ranges and selected the flavour holds for the smaller ones.  MIJPEG_NO_NARROW12: A-B runs.
static bool narrow12_colour(const mijpeg_info &f)
{
  static const bool off = getenv("MIJPEG_NO_NARROW12") != nullptr;
  if (off || f.precision != 12 || f.components != 3) return false;
  const int64_t ry = f.range_max[0], rc = std::max(f.range_max[1], f.range_max[2]);
  if (ry <= 0 || rc < 0) return false;
  const int64_t sum = ((402 * ry + 99) / 100 + 2 + 32776) * 8192 + 14516 * ((402 * rc + 99) / 100 + 4);
  return sum < ((int64_t)1 << 31);
}

// 12-bit 4:4:4 frames: the same bounds (no filter between the transforms and the colour stage)
static bool use_fused444_12(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F444_12") != nullptr; // A-B comparisons
  if (off || !is_444(f) || !f.ycbcr || f.xt || f.precision != 12 ||
      (b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_NO_COLOR_TRANSFORM | MIJPEG_FLAG_FORCE_SAFE)) || !fits32(b))
    return false;
  if (f.range_max[0] <= 0 || f.range_max[0] >= 49152 || f.range_max[1] >= 45056 || f.range_max[2] >= 45056) return false;
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < 64; i++)
      if (f.quant[f.quant_index[c]][i] > 2047) return false;
  return true;
}

// 12-bit 4:2:2 frames: the same bounds again (the horizontal filter weighs samples below 2^18 with 4 in total)
static bool use_fused422_12(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F422_12") != nullptr; // A-B comparisons
  if (off || !is_422(f) || !f.ycbcr || f.xt || f.precision != 12 ||
      (b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_NO_COLOR_TRANSFORM | MIJPEG_FLAG_FORCE_SAFE)) || !fits32(b))
    return false;
  if (f.range_max[0] <= 0 || f.range_max[0] >= 49152 || f.range_max[1] >= 45056 || f.range_max[2] >= 45056) return false;
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < 64; i++)
      if (f.quant[f.quant_index[c]][i] > 2047) return false;
  return true;
}

// 12-bit single-component frames: the butterflies' bound alone
static bool use_fused1_12(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F1_12") != nullptr; // A-B comparisons
  if (off || f.components != 1 || f.xt || f.precision != 12 || (b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_FORCE_SAFE)) || !fits32(b)) return false;
  if (f.range_max[0] <= 0 || f.range_max[0] >= 49152) return false;
  for (int i = 0; i < 64; i++)
    if (f.quant[f.quant_index[0]][i] > 2047) return false;
  return true;
}

// 12-bit frames of the other layouts (fused_tile_kernel): the same bounds, the chroma one for every component -- 32-bit
// butterflies and colour products as in use_fused420_12; the upsampling filters weigh two samples (< 2^18 each with the level
// shift) with at most 8 in total, far inside the 24-bit operands and 32-bit sums of the kernel's fast flavour
static bool tile_fast12(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_TILE_FAST12") != nullptr; // A-B comparisons
  if (off || f.xt || f.precision != 12 || f.coef_wide || (b->flags & MIJPEG_FLAG_FORCE_SAFE) || f.range_max[0] <= 0) return false;
  for (int c = 0; c < f.components; c++) {
    if (f.range_max[c] >= 45056) return false;
    for (int i = 0; i < 64; i++)
      if (f.quant[f.quant_index[c]][i] > 2047) return false;
  }
  return true;
}

// the packed flavour filters (Cb, Cr) pairs in 16 bits: every chroma sample * 16 is bounded by 4 * range_max, and the
// filter sums a + 3 b + r by four times that
static bool use_fused420p(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F420P") != nullptr; // tuning / A-B comparisons
  return !off && use_fused420(b) && fast_ok(b) && f.range_max[1] < 2047 && f.range_max[2] < 2047;
}

// ... and where the first-pass results of every transform fit 16 bits the second pass runs on v_dot2 as well (idct_columns_dot2 in
// kernels.hip has the bound: sum |c| q <= 1476 per block).  MIJPEG_FLAG_FORCE_DOT2 (testing): whatever the range check says.
static bool use_dot2_pass(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_DOT2") != nullptr; // A-B comparisons
  if (off || b->quant_dev) return false;
  if (b->flags & MIJPEG_FLAG_FORCE_DOT2) return true;
  return f.range_max[0] <= 1476 && f.range_max[1] <= 1476 && f.range_max[2] <= 1476;
}

// JPEG XT profile C in the shape the fused kernel covers: 8-bit 4:2:0 legacy frame and 12-bit 4:4:4 residual frame without
// hidden bits, L transformation on, both frames within the range the fast transforms are exact for
// JPEG XT: the L transformation in force for this launch.  A request without colour transformation (the command line's -c)
// replaces the STANDARD YCbCr transformation by the identity and leaves everything else of the merge alone
// (colortrafo/colortransformerfactory.cpp:231-232: `if (ltrafo == YCbCr && disabletorgb) ltrafo = Identity`)
static bool xt_ltrafo_ycbcr(const mijpeg_batch *b)
{
  const mijpeg_xt_params &x = *b->xt;
  return x.ltrafo_ycbcr && !((b->flags & MIJPEG_FLAG_NO_COLOR_TRANSFORM) && x.ltrafo_standard);
}

static bool use_fusedxt(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  if (!f.xt || !b->xt || !is_420(f) || f.precision != 8 || (b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_FORCE_SAFE)) || !fits32(b)) return false;
  // the legacy frame's range check (fast_arith itself is never set for XT frames: the generic kernels run SAFE on them)
  for (int c = 0; c < 3; c++) {
    if (f.range_max[c] >= 16384) return false;
    for (int i = 0; i < 64; i++)
      if (f.quant[f.quant_index[c]][i] > 2047) return false;
  }
  static const bool off = getenv("MIJPEG_NO_FUSEDXT") != nullptr; // A-B comparisons
  if (off) return false;
  const mijpeg_xt_params &x = *b->xt;
  const mijpeg_info &r = x.residual;
  if (x.general) return false; // free-form matrices, table gathers, DCT bypass: xt_merge_general_kernel
  if (x.no_residual) return false; // (a legacy codestream without its EOI: the unfused merge kernels know how to merge nothing)
  // hidden bits in the RESIDUAL frame (-rR n: 13..16-bit samples, int32 coefficients) have a kernel of their own
  // (fusedxtw420_kernel); hidden bits in the legacy frame change its precision and stay on the three-kernel path
  if (x.hidden_bits || x.residual_hidden_bits < 0 || x.residual_hidden_bits > 4 || (x.residual_wide != 0) != (x.residual_hidden_bits > 0) ||
      x.ltable_entries != 256 || !xt_ltrafo_ycbcr(b) || r.precision != 12 || r.components != 3 || x.out_max != 65535 || x.out_shift != 32768)
    return false;
  static const bool no_wide = getenv("MIJPEG_NO_FUSEDXTW") != nullptr; // A-B comparisons
  if (x.residual_hidden_bits && no_wide) return false;
  for (int c = 0; c < 3; c++) {
    if (r.subx[c] != 1 || r.suby[c] != 1 || r.blocks_w[c] != r.blocks_w[0] || r.blocks_h[c] != r.blocks_h[0] || r.range_max[c] >= 65536) return false;
    for (int i = 0; i < 64; i++)
      if (r.quant[r.quant_index[c]][i] > 2047) return false;
  }
  return r.width == f.width && r.height == f.height;
}

// single-component frames (and single components without upsampling): samples travel as packed int16
static bool use_fused1(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F1") != nullptr; // A-B comparisons
  return !off && f.components == 1 && !f.xt && f.precision == 8 && !(b->flags & MIJPEG_FLAG_FORCE_GENERIC) && fast_ok(b) && f.range_max[0] < 8190 &&
         fits32(b);
}

// fused 4:2:2 / 4:4:0: chroma samples travel through LDS as int16 pairs (range_max < 8190, the fused 4:4:4 kernel's bound);
// below the packed 4:2:0 flavour's bound (2047) the filter runs on the pairs, between the two on unpacked 32-bit values
static bool chroma_packed(const mijpeg_info &f) { return f.range_max[1] < 2047 && f.range_max[2] < 2047; }
static bool use_fused422(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F422") != nullptr; // A-B comparisons
  return !off && is_422(f) && f.ycbcr && !f.xt && f.precision == 8 && !(b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_NO_COLOR_TRANSFORM)) && fast_ok(b) &&
         f.range_max[1] < 8190 && f.range_max[2] < 8190 && fits32(b);
}

// fused 4:1:1: int16 pairs in LDS, 32-bit four-fold filter
static bool use_fused411(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F411") != nullptr; // A-B comparisons
  return !off && is_411(f) && f.ycbcr && !f.xt && f.precision == 8 && !(b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_NO_COLOR_TRANSFORM)) && fast_ok(b) &&
         f.range_max[1] < 8190 && f.range_max[2] < 8190 && fits32(b);
}

// fused 4:4:0 (what a losslessly rotated 4:2:2 picture is): the vertical half of the packed filter, same bound
static bool use_fused440(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_F440") != nullptr; // A-B comparisons
  return !off && is_440(f) && f.ycbcr && !f.xt && f.precision == 8 && !(b->flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_NO_COLOR_TRANSFORM)) && fast_ok(b) &&
         f.range_max[1] < 8190 && f.range_max[2] < 8190 && fits32(b);
}

// every component 1 x 1, three or four of them, 8 bit, no colour transformation, fast arithmetic: fused_flat_kernel
// (CMYK; RGB stored as such -- Adobe transform 0, a merging specification with the identity L transformation, the caller's
// MIJPEG_FLAG_NO_COLOR_TRANSFORM on a 4:4:4 frame)
static bool use_fused_flat(const mijpeg_batch *b)
{
  const mijpeg_info &f = b->info;
  static const bool off = getenv("MIJPEG_NO_FUSED_FLAT") != nullptr; // A-B measurements
  if (off || f.xt || f.precision != 8 || f.coef_wide || b->quant_dev || (f.components != 3 && f.components != 4)) return false;
  if (b->flags & MIJPEG_FLAG_FORCE_GENERIC) return false;
  if (f.components == 3 && f.ycbcr && !(b->flags & MIJPEG_FLAG_NO_COLOR_TRANSFORM)) return false;
  for (int c = 0; c < f.components; c++)
    if (f.subx[c] != 1 || f.suby[c] != 1 || f.blocks_w[c] != f.blocks_w[0] || f.blocks_h[c] != f.blocks_h[0]) return false;
  return fast_ok(b) && fits32(b) && !dnl_row_missing(f);
}

const char *mijpeg_kernel_name(const mijpeg_batch *b)
try {
  if (!b) return "";
  if (use_fusedxt(b)) return b->xt->residual_hidden_bits ? "fusedxtw420_kernel" : "fusedxt420_kernel";
  if (use_fused420p(b)) return "fused420p_kernel";
  if (use_fused422(b)) return chroma_packed(b->info) ? "fused422_kernel" : "fused422_kernel<wide>";
  if (use_fused440(b)) return chroma_packed(b->info) ? "fused440_kernel" : "fused440_kernel<wide>";
  if (use_fused411(b)) return "fused411_kernel";
  if (use_fused1(b)) return "fused1_kernel";
  if (use_fused420_12(b)) return narrow12_colour(b->info) ? "fused420_kernel<12>/narrow" : "fused420_kernel<12>";
  if (use_fused1_12(b)) return "fused1_kernel<12>";
  if (use_fused444_12(b)) return narrow12_colour(b->info) ? "fused444_12_kernel/narrow" : "fused444_12_kernel";
  if (use_fused422_12(b)) return narrow12_colour(b->info) ? "fused422_12_kernel/narrow" : "fused422_12_kernel";
  if (b->info.coef_wide) return "idct_planes_long_kernel+upsample_color_kernel";
  if (use_fused420(b)) return "fused420_kernel";
  if (use_fused444(b)) return "fused444_kernel";
  if (b->info.xt && b->info.components == 1) return "idct_planes_kernel+xt_merge1_kernel";
  if (b->info.xt) return b->xt && b->xt->general ? "idct_planes_kernel+xt_merge_general_kernel" : "idct_planes_kernel+xt_merge_kernel";
  if (use_fused_flat(b)) return "fused_flat_kernel";
  if (b->quant_dev || (b->flags & MIJPEG_FLAG_FORCE_GENERIC) || getenv("MIJPEG_NO_FUSED_TILE") || dnl_row_missing(b->info)) return "idct_planes_kernel+upsample_color_kernel";
  return "fused_tile_kernel";
} catch (...) { (void)boundary_catch(nullptr, "mijpeg_kernel_name"); return nullptr; }

static const size_t LUT_BYTES = 3 * 4096 * sizeof(int32_t);

// JPEG XT with real Q / R2 tables (mijpeg_xt_params.general): they travel in the workspace behind everything else
static size_t xt_table_bytes(const mijpeg_batch *b)
{
  if (!b->info.xt || !b->xt || !b->xt->general) return 0;
  size_t n = 0;
  for (int c = 0; c < 3; c++) {
    if (b->xt->qtable[c]) n += (size_t)b->xt->qtable_entries * sizeof(int32_t);
    if (b->xt->r2table[c]) n += ((size_t)(b->xt->out_max + 1) << 4) * sizeof(int32_t);
  }
  return n;
}

// per-frame tables (quant_dev) are expanded to the transforms' operands (deltas << 4, int32) in the workspace
static size_t expanded_tables_bytes(const mijpeg_batch *b) { return b->quant_dev ? (size_t)b->frames * 4 * 64 * sizeof(int32_t) : 0; }

size_t mijpeg_workspace_bytes(const mijpeg_batch *b)
try {
  if (!b) return 0;
  if (use_fused420(b) || use_fused444(b) || use_fused422(b) || use_fused440(b) || use_fused411(b) || use_fused1(b) || use_fused420_12(b) || use_fused1_12(b) || use_fused444_12(b) || use_fused422_12(b)) return expanded_tables_bytes(b);
  if (use_fusedxt(b)) return LUT_BYTES;
  // [LUT_BYTES: L lookup tables (JPEG XT, up to 3 x 4096 entries)] [per frame: int32 sample planes, one sample per
  // coefficient: coef_count of them, fewer when the residual planes hold 32-bit coefficients] [expanded per-frame tables]
  return LUT_BYTES + (size_t)b->info.coef_count * sizeof(int32_t) * (size_t)b->frames + expanded_tables_bytes(b) + xt_table_bytes(b);
} catch (...) { (void)boundary_catch(nullptr, "mijpeg_workspace_bytes"); return 0; }

// What a rectangle request that does not show the plain picture adds to a launch (request_model.hpp; GenericArgs::rowmap ...)
struct RequestExtra {
  const int32_t *rowmap_dev;
  int32_t rowmap_stride;
  int32_t corner_x, corner_y, y_base, y_count;
  int32_t wstart[MAXP], wlimit[MAXP]; // per plane (JPEG XT: legacy planes, then residual planes)
  int32_t ycc;
};
static int launch_reconstruct_ex(const mijpeg_batch *b, void *stream, const RequestExtra *rx);

int mijpeg_launch_reconstruct(const mijpeg_batch *b, void *stream)
try {
  return launch_reconstruct_ex(b, stream, nullptr);
} catch (...) { return boundary_catch(nullptr, "mijpeg_launch_reconstruct"); }

static int launch_reconstruct_ex(const mijpeg_batch *b, void *stream, const RequestExtra *rx)
{
  if (!b || !b->coef_dev || !b->out_dev || b->frames < 1) return MIJPEG_ERR_INVALID_PARAMETER;
  if (rx && !(b->flags & MIJPEG_FLAG_FORCE_GENERIC)) return MIJPEG_ERR_INVALID_PARAMETER;
  if (b->quant_dev && b->info.xt) return MIJPEG_ERR_OPERATION_UNIMPLEMENTED; // per-frame tables: plain JPEG only
  const mijpeg_info &f = b->info;
  if ((f.precision != 8 && f.precision != 12) || f.components < 1 || f.components > 4) return MIJPEG_ERR_OPERATION_UNIMPLEMENTED;
  if (f.xt && (!b->xt || (f.components != 3 && f.components != 1))) return MIJPEG_ERR_MISSING_PARAMETER; // (one component: grey scale with a residual)
  if (f.coef_wide && (f.xt || b->quant_dev)) return MIJPEG_ERR_INVALID_PARAMETER; // int32 planes: single plain JPEG frames only
  const bool fast = fast_ok(b) && !f.coef_wide;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const bool f444 = use_fused444(b), fxt = use_fusedxt(b), f422 = use_fused422(b), f440 = use_fused440(b), f411 = use_fused411(b), f1 = use_fused1(b);
  const bool f420_12 = use_fused420_12(b), f1_12 = use_fused1_12(b), f444_12 = use_fused444_12(b), f422_12 = use_fused422_12(b);
  if (fxt && (!b->workspace || b->workspace_bytes < LUT_BYTES)) return MIJPEG_ERR_MISSING_PARAMETER;
  const int32_t *qdev = nullptr;
  if (b->quant_dev) {
    const size_t need = mijpeg_workspace_bytes(b);
    if (!b->workspace || b->workspace_bytes < need) return MIJPEG_ERR_MISSING_PARAMETER;
    int32_t *dst = (int32_t *)((char *)b->workspace + (need - expanded_tables_bytes(b) - xt_table_bytes(b)));
    if (launch_expand_deltas(b->quant_dev, dst, b->frames, s)) return MIJPEG_ERR_DEVICE;
    qdev = dst;
  }
  if (use_fused420(b) || f444 || fxt || f422 || f440 || f411 || f1 || f420_12 || f1_12 || f444_12 || f422_12) {
    FusedXtArgs xa;
    memset(&xa, 0, sizeof(xa));
    Fused420Args &a = xa.base;
    a.coef = b->coef_dev;
    a.coef_frame_stride = b->coef_frame_stride;
    a.off_y = f.coef_offset[0];
    a.off_cb = f.coef_offset[1];
    a.off_cr = f.coef_offset[2];
    a.out = b->out_dev;
    a.out_frame_stride = b->out_frame_stride;
    a.row_stride = b->out_row_stride;
    a.width = f.width;
    a.height = f.height;
    a.bw_y = f.blocks_w[0];
    a.bh_y = f.blocks_h[0];
    a.bw_c = f.blocks_w[1];
    a.bh_c = f.blocks_h[1];
    a.cw = f440 ? f.width : f411 ? (f.width + 3) / 4 : (f.width + 1) / 2;
    a.ch = (f422 || f411 || f422_12) ? f.height : (f.height + 1) / 2;
    // DNL frames: the reference's upsamplers never learnt the height (upsampling/upsamplerbase.cpp:61-75), their line buffers
    // have no bottom edge: below the last chroma line comes what the block rows hold (the padding of the last one, then the
    // MCU row the first scan created behind the picture: the store has it, include/mijpeg.h) instead of that line again
    if (f.dnl && !(f422 || f411 || f422_12) && f.components > 1) a.ch = a.bh_c * 8;
    a.tiles_x = (f.width + 127) / 128;
    a.tiles_y = (f.height + 127) / 128;
    a.frames = b->frames;
    for (int c = 0; c < 3; c++)
      fill_deltas(a.q[c], f.quant[f.quant_index[c]]);
    a.qdev = qdev;
    if (fxt) {
      const mijpeg_xt_params &x = *b->xt;
      const mijpeg_info &r = x.residual;
      for (int c = 0; c < 3; c++) {
        xa.ext.off_r[c] = r.coef_offset[c];
        for (int i = 0; i < 64; i++) xa.ext.rq[c][i] = (int32_t)r.quant[r.quant_index[c]][i] << 4;
        if (hipMemcpyAsync((int32_t *)b->workspace + (size_t)c * 256, x.ltable[c], 256 * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess)
          return MIJPEG_ERR_DEVICE;
      }
      xa.ext.bw_r = r.blocks_w[0];
      xa.ext.bh_r = r.blocks_h[0];
      xa.ext.ltable = (const int32_t *)b->workspace;
      xa.ext.rtrafo_ycbcr = x.rtrafo_ycbcr;
      xa.ext.is_float = x.is_float;
      xa.ext.out_max = x.out_max;
      xa.ext.out_shift = x.out_shift;
      xa.ext.rprecision = r.precision + x.residual_hidden_bits;
      // (the two-wave flavour of the hidden-bit kernel keeps the luma block as int16: sample * 16 + 2056 with |sample * 16| <= 4 sum |c| q)
      xa.luma_fits16 = f.range_max[0] < 7600 ? 1 : 0;
      rc = launch_fusedxt420(xa, s);
    } else
      rc = f420_12 ? launch_fused420_12(a, narrow12_colour(f), s) : f444_12 ? launch_fused444_12(a, narrow12_colour(f), s) : f422_12 ? launch_fused422_12(a, narrow12_colour(f), s) : f1_12 ? launch_fused1_12(a, s) : f1 ? launch_fused1(a, s) : f444 ? launch_fused444(a, s) : f422 ? launch_fused422(a, !chroma_packed(f), s) : f440 ? launch_fused440(a, !chroma_packed(f), s) : f411 ? launch_fused411(a, s) : use_fused420p(b) ? launch_fused420p(a, use_dot2_pass(b), s) : launch_fused420(a, fast, s);
  } else {
    if (!b->workspace || b->workspace_bytes < mijpeg_workspace_bytes(b)) return MIJPEG_ERR_MISSING_PARAMETER;
    GenericArgs a;
    memset(&a, 0, sizeof(a));
    a.coef = b->coef_dev;
    a.coef_frame_stride = b->coef_frame_stride;
    a.samples = (int32_t *)((char *)b->workspace + LUT_BYTES);
    a.sample_frame_stride = f.coef_count;
    a.out = b->out_dev;
    a.out_frame_stride = b->out_frame_stride;
    a.row_stride = b->out_row_stride;
    a.width = f.width;
    a.height = f.height;
    a.ncomp = f.components;
    a.ycbcr = (f.ycbcr && !(b->flags & MIJPEG_FLAG_NO_COLOR_TRANSFORM)) ? 1 : 0;
    a.frames = b->frames;
    a.qdev = qdev;
    a.nplanes = f.components;
    a.sample_bytes = f.xt ? (b->xt->out_max > 255 ? 2 : 1) : f.precision > 8 ? 2 : 1;
    a.maxval = (1 << f.precision) - 1;
    a.dcshift = (1 << (f.precision - 1)) << 4;
    int64_t sample_off = 0;
    auto plane = [&](int p, const mijpeg_info &g, int c, int precision) {
      a.coef_off[p] = g.coef_offset[c];
      a.sample_off[p] = sample_off;
      sample_off += (int64_t)g.blocks_w[c] * g.blocks_h[c] * 64;
      a.bw[p] = g.blocks_w[c];
      a.bh[p] = g.blocks_h[c];
      a.subx[p] = g.subx[c];
      a.suby[p] = g.suby[c];
      a.cw[p] = (g.width + g.subx[c] - 1) / g.subx[c];
      a.ch[p] = (g.height + g.suby[c] - 1) / g.suby[c];
      if (g.dnl && g.suby[c] > 1) { // no bottom edge, see above; rows nobody created are NULL: zeros
        if ((a.ch[p] & 7) == 0 && g.rows[c] <= (a.ch[p] >> 3)) a.zero_from[p] = g.rows[c];
        a.ch[p] = g.blocks_h[c] * 8;
      }
      a.dcoff[p] = (1 << (precision - 1)) << 7;
      fill_deltas(a.q[p], g.quant[g.quant_index[c]]);
    };
    // JPEG XT frames reconstruct at their precision plus the bits that travelled in hidden refinement scans
    // (Frame::HiddenPrecisionOf, marker/frame.cpp:368-373)
    const int lprec = f.precision + (f.xt ? b->xt->hidden_bits : 0);
    for (int c = 0; c < f.components; c++) plane(c, f, c, lprec);
    if (f.coef_wide) { a.wide_first = 0; a.wide_count = f.components; a.wide_long = 1; }
    // int16 sample planes between the two kernels: |sample * 16| <= 2048 (level shift) + 4 * range_max must fit 16 bits
    static const bool int32_planes = getenv("MIJPEG_GENERIC_INT32") != nullptr; // A-B comparisons
    a.narrow = fast && !f.xt && f.precision == 8 && !f.coef_wide && !int32_planes;
    for (int c = 0; c < f.components && a.narrow; c++)
      if (f.range_max[c] >= 7600) a.narrow = 0;
    a.maxval = (1 << lprec) - 1;
    a.dcshift = (1 << (lprec - 1)) << 4;
    if (f.xt) {
      const mijpeg_xt_params &x = *b->xt;
      const int rprec = x.residual.precision + x.residual_hidden_bits;
      // (a parameter block filled in before the lossless flavours existed has zeros there: with clamping that means four bits)
      const int xrbits = (x.rbits == 0 && x.clamp) ? 4 : x.rbits;
      if (x.hidden_bits < 0 || x.hidden_bits > 4 || x.residual_hidden_bits < 0 || x.residual_hidden_bits > 4 || rprec - (x.rct ? 1 : 0) > 16 ||
          x.ltable_entries != (256 << x.hidden_bits) || (x.residual_wide != 0) != (x.residual_hidden_bits > 0 || x.residual.precision > 12) ||
          (xrbits != 4 && !(x.general && x.rdct_bypass)) || (x.rct && (x.clamp || xrbits != 1)) || (!x.clamp && !x.general))
        return MIJPEG_ERR_INVALID_PARAMETER;
      // the flavours without clamping (RCT, lossless identity) index their Q tables directly: a caller-made block without them is refused
      if ((x.rct || !x.clamp) && !x.no_residual)
        for (int c = 0; c < x.residual.components && c < 3; c++)
          if (!x.qtable[c]) return MIJPEG_ERR_INVALID_PARAMETER;
      for (int c = 0; c < x.residual.components && c < 3; c++) plane(3 + c, x.residual, c, rprec); // (one component: planes 4, 5 stay empty)
      if (!x.residual.components) // (no residual frame at all -- a specification without a residual codestream: the merge reads nothing there)
        for (int pn = 3; pn < 6; pn++) a.subx[pn] = a.suby[pn] = 1;
      // int32 planes: beyond 12 bits (hidden bits included) the reference transforms with IDCT<4,QUAD>, up to 12 with the LONG
      // flavour like every other frame (codestream/tables.cpp:1876-1891) -- the same numbers until a damaged scan leaves a
      // coefficient that overflows 32 bits on the way (an 8-bit alpha residual with one hidden bit and 52 241 in a block:
      // tools/xt_gpu_damage_campaign.py, seed 2002)
      if (x.residual_wide) { a.wide_first = 3; a.wide_count = 3; a.wide_long = rprec <= 12 ? 1 : 0; }
      a.ltable_entries = x.ltable_entries;
      a.nplanes = 6;
      a.xt = 1;
      a.ycbcr = xt_ltrafo_ycbcr(b) ? 1 : 0; // the L transformation of the merging specification, or the identity the -c switch puts in its place
      a.rtrafo_ycbcr = x.rtrafo_ycbcr;
      a.out_shift = x.out_shift;
      a.out_max = x.out_max;
      a.is_float = x.is_float;
      a.rprecision = rprec;
      a.xt_no_residual = x.no_residual;
      a.xt_rct = x.rct;
      a.xt_noclamp = x.clamp ? 0 : 1;
      a.xt_rbits = x.residual.components ? xrbits : 4;
      a.legacy32 = lprec == 8 && f.range_max[0] < 16384 && f.range_max[1] < 16384 && f.range_max[2] < 16384 &&
                   !(b->flags & MIJPEG_FLAG_FORCE_SAFE);
      a.ltable = (const int32_t *)b->workspace;
      for (int c = 0; c < 3; c++)
        if (hipMemcpyAsync((int32_t *)b->workspace + (size_t)c * x.ltable_entries, x.ltable[c], (size_t)x.ltable_entries * sizeof(int32_t),
                           hipMemcpyHostToDevice, s) != hipSuccess)
          return MIJPEG_ERR_DEVICE;
      if (x.general) {
        if (x.residual.components && x.qtable_entries != (1 << (rprec - (xrbits == 1) + xrbits))) return MIJPEG_ERR_INVALID_PARAMETER; // (no residual frame: no Q tables)
        a.xt_general = 1;
        a.rbypass = x.rdct_bypass;
        a.rnoise = x.noise_shaping;
        a.rdcshift = (1 << rprec) >> 1;
        memcpy(a.lmat, x.lmat, sizeof(a.lmat));
        memcpy(a.rmat, x.rmat, sizeof(a.rmat));
        memcpy(a.cmat, x.cmat, sizeof(a.cmat));
        char *tp = (char *)b->workspace + (mijpeg_workspace_bytes(b) - xt_table_bytes(b));
        for (int c = 0; c < 3; c++) {
          // only the highest-frequency delta is used, with the colour bits folded in (residualblockhelper.cpp:351-364)
          // (m_usQuantization is a UWORD: deltas >= 4096 wrap; shifted where the path has more than one fractional bit)
          a.rquant63[c] = xrbits > 1 ? ((int32_t)x.residual.quant[x.residual.quant_index[c]][63] << xrbits) & 0xffff : (int32_t)x.residual.quant[x.residual.quant_index[c]][63];
          // (components that share a table share its copy)
          for (int j = 0; j < c; j++) {
            if (x.qtable[c] && x.qtable[j] == x.qtable[c]) a.qlut[c] = a.qlut[j];
            if (x.r2table[c] && x.r2table[j] == x.r2table[c]) a.r2lut[c] = a.r2lut[j];
          }
          if (x.qtable[c] && !a.qlut[c]) {
            const size_t n = (size_t)x.qtable_entries * sizeof(int32_t);
            if (hipMemcpyAsync(tp, x.qtable[c], n, hipMemcpyHostToDevice, s) != hipSuccess) return MIJPEG_ERR_DEVICE;
            a.qlut[c] = (const int32_t *)tp;
            tp += n;
          }
          if (x.r2table[c] && !a.r2lut[c]) {
            const size_t n = ((size_t)(x.out_max + 1) << 4) * sizeof(int32_t);
            if (hipMemcpyAsync(tp, x.r2table[c], n, hipMemcpyHostToDevice, s) != hipSuccess) return MIJPEG_ERR_DEVICE;
            a.r2lut[c] = (const int32_t *)tp;
            tp += n;
          }
        }
      }
    }
    if (rx) {
      a.rowmap = rx->rowmap_dev;
      a.rowmap_stride = rx->rowmap_stride;
      a.request = 1;
      a.req_x0 = rx->corner_x;
      a.req_y0 = rx->corner_y;
      a.y_base = rx->y_base;
      a.y_count = rx->y_count;
      for (int c = 0; c < a.nplanes && c < MAXP; c++) {
        a.wstart[c] = rx->wstart[c];
        a.wlimit[c] = rx->wlimit[c];
      }
      if (!f.xt) a.ycbcr = rx->ycc; // the colour transformer the first request built (colortransformerfactory.cpp:220-221)
    }
    // plain JPEG frames of any layout go through LDS in one pass (fused_tile_kernel); the pair with its sample planes in HBM
    // stays for JPEG XT, int32 coefficient planes, per-frame tables in device memory, rectangle requests and MIJPEG_FLAG_FORCE_GENERIC
    static const bool no_tile = getenv("MIJPEG_NO_FUSED_TILE") != nullptr; // A-B measurements
    const bool tile = !rx && !f.xt && !f.coef_wide && !qdev && !(b->flags & MIJPEG_FLAG_FORCE_GENERIC) && !no_tile && !dnl_row_missing(f);
    rc = !rx && use_fused_flat(b) ? launch_fused_flat(a, s) : -1;
    if (rc == -1) rc = tile ? launch_fused_tile(a, fast || tile_fast12(b), s) : -1;
    if (rc == -1) rc = launch_generic(a, fast, s);
  }
  return rc ? MIJPEG_ERR_DEVICE : MIJPEG_OK;
}

// ------------------------------------------------------------------------------------------------
// encoder direction of the block pipeline
// ------------------------------------------------------------------------------------------------
int mijpeg_frame_layout(mijpeg_info *f)
try {
  if (!f || f->width < 1 || f->height < 1 || f->width > 65535 || f->height > 65535 || f->components < 1 || f->components > MIJPEG_MAX_COMPONENTS)
    return MIJPEG_ERR_INVALID_PARAMETER;
  int hmax = 1, vmax = 1;
  for (int c = 0; c < f->components; c++) {
    if (f->hsamp[c] < 1 || f->hsamp[c] > 4 || f->vsamp[c] < 1 || f->vsamp[c] > 4 || f->quant_index[c] < 0 || f->quant_index[c] > 3)
      return MIJPEG_ERR_INVALID_PARAMETER;
    hmax = std::max(hmax, f->hsamp[c]);
    vmax = std::max(vmax, f->vsamp[c]);
  }
  f->mcus_x = (f->width + 8 * hmax - 1) / (8 * hmax);
  f->mcus_y = (f->height + 8 * vmax - 1) / (8 * vmax);
  int64_t off = 0;
  for (int c = 0; c < f->components; c++) {
    if (hmax % f->hsamp[c] || vmax % f->vsamp[c]) return MIJPEG_ERR_INVALID_PARAMETER; // fractional subsampling factors
    f->subx[c] = hmax / f->hsamp[c];
    f->suby[c] = vmax / f->vsamp[c];
    f->blocks_w[c] = f->mcus_x * f->hsamp[c];
    f->blocks_h[c] = f->mcus_y * f->vsamp[c];
    f->coef_offset[c] = off;
    off += (int64_t)f->blocks_w[c] * f->blocks_h[c] * 64;
  }
  f->coef_count = off;
  f->sample_bytes = 1;
  return MIJPEG_OK;
} catch (...) { return boundary_catch(nullptr, "mijpeg_frame_layout"); }

int mijpeg_launch_forward(const mijpeg_forward_batch *b, void *stream)
try {
  if (!b || !b->pixels_dev || !b->coef_dev || b->frames < 1) return MIJPEG_ERR_INVALID_PARAMETER;
  const mijpeg_info &f = b->info;
  if (f.precision != 8 || (f.components != 1 && f.components != 3) || f.xt) return MIJPEG_ERR_OPERATION_UNIMPLEMENTED;
  ForwardArgs a;
  memset(&a, 0, sizeof(a));
  a.pixels = b->pixels_dev;
  a.pixel_frame_stride = b->pixel_frame_stride;
  a.pixel_row_stride = b->pixel_row_stride;
  a.coef = b->coef_dev;
  a.coef_frame_stride = b->coef_frame_stride;
  a.width = f.width;
  a.height = f.height;
  a.ncomp = f.components;
  a.ycbcr = f.ycbcr;
  a.frames = b->frames;
  const bool dword_lines = (((uintptr_t)b->pixels_dev | (uintptr_t)b->pixel_frame_stride | (uintptr_t)b->pixel_row_stride) & 3) == 0;
  uint64_t blocks = 0;
  for (int c = 0; c < f.components; c++) {
    if (f.subx[c] < 1 || f.suby[c] < 1 || f.blocks_w[c] < 1 || f.blocks_h[c] < 1) return MIJPEG_ERR_INVALID_PARAMETER;
    a.subx[c] = f.subx[c];
    a.suby[c] = f.suby[c];
    a.bw[c] = f.blocks_w[c];
    a.bh[c] = f.blocks_h[c];
    a.nbx[c] = ((f.width + f.subx[c] - 1) / f.subx[c] + 7) >> 3;
    a.nby[c] = ((f.height + f.suby[c] - 1) / f.suby[c] + 7) >> 3;
    a.coef_off[c] = f.coef_offset[c];
    a.fast[c] = dword_lines && f.components == 3 && f.ycbcr && f.subx[c] <= 2 && f.suby[c] <= 2 && !getenv("MIJPEG_FORWARD_SLOW");
    a.fast_nbx[c] = f.width / (8 * f.subx[c]);
    a.fast_nby[c] = f.height / (8 * f.suby[c]);
    a.first_block[c] = (uint32_t)blocks;
    blocks += (uint64_t)f.blocks_w[c] * f.blocks_h[c];
    for (int i = 0; i < 64; i++) {
      const uint16_t delta = f.quant[f.quant_index[c]][i];
      if (delta == 0) return MIJPEG_ERR_INVALID_PARAMETER;
      // LONG(FLOAT(1L << QUANTIZER_BITS) / delta + 0.5), dct/idct.cpp:106: a single precision quotient
      volatile float q = (float)(1L << 30) / (float)delta;
      a.invq[c][i] = (int32_t)((double)q + 0.5);
    }
  }
  if (blocks > 0xffffffffull) return MIJPEG_ERR_INVALID_PARAMETER;
  a.first_block[f.components] = (uint32_t)blocks;
  if (a.fast[0] && a.fast[1] && a.fast[2] && f.subx[0] == 1 && f.suby[0] == 1 && f.subx[1] == 2 && f.suby[1] == 2 && f.subx[2] == 2 && f.suby[2] == 2 &&
      f.width >= 128 && f.height >= 128 && !getenv("MIJPEG_FORWARD_NO_TILES")) {
    a.tiled420 = 1;
    const int tx = f.width >> 7, ty = f.height >> 7;
    a.tile_nbx[0] = tx * 16; a.tile_nby[0] = ty * 16;
    for (int c = 1; c < 3; c++) { a.tile_nbx[c] = tx * 8; a.tile_nby[c] = ty * 8; }
  }
  return launch_forward(a, (hipStream_t)stream) ? MIJPEG_ERR_DEVICE : MIJPEG_OK;
} catch (...) { return boundary_catch(nullptr, "mijpeg_launch_forward"); }

void mijpeg_quality_tables(int quality, uint16_t luma[64], uint16_t chroma[64])
try {
  // ISO/IEC 10918-1 Annex K.1 / K.2 matrices, natural order
  static const uint8_t K1[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,  69,  56,
                                 14, 17, 22, 29, 51,  87,  80,  62,  18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
  static const uint8_t K2[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
  quality = std::min(100, std::max(1, quality));
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2; // quantization.cpp:296-299
  for (int j = 0; j < 64; j++) {
    luma[j] = (uint16_t)std::min(255, std::max(1, (K1[j] * scale + 50) / 100)); // :411, :443-466
    chroma[j] = (uint16_t)std::min(255, std::max(1, (K2[j] * scale + 50) / 100));
  }
} catch (...) { (void)boundary_catch(nullptr, "mijpeg_quality_tables"); }

// Entropy coding of one frame's coefficient planes on the device (hencode.hip) and download of the finished stream, as a
// job of three stages with a host synchronisation in front of the second and the third (the byte counts the next stage
// sizes its buffers and copies with come from the device).  Two jobs on two streams with two sets of buffers overlap:
// mijpeg_encode_batch_device keeps the next frame's first stage in flight while it waits for the current frame.
struct HencJob {
  mijpeg_decoder *d = nullptr;
  const mijpeg_info *f = nullptr;
  int slot = 0, restart_interval = 0;
  hipStream_t stream = nullptr;
  HencArgs a;
  EncTables tabs;
  uint64_t *readback = nullptr; // pinned: [0] plain bytes, [1] 0xFF bytes
  uint64_t *scratch = nullptr;
  uint32_t chunks = 0;
  uint8_t *result = nullptr;
  size_t head_size = 0, ecs = 0;
  std::vector<uint8_t> head;

  static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

  int upload_tables()
  {
    HencTables *h = (HencTables *)((uint8_t *)d->henc_host + 64 + (size_t)slot * sizeof(HencTables)); // pinned, one per slot
    memset(h, 0, sizeof(*h));
    for (int t = 0; t < 2; t++) {
      for (int k = 0; k < 16; k++) { h->dc_code[t][k] = tabs.dc[t].code[k]; h->dc_len[t][k] = tabs.dc[t].len[k]; }
      for (int k = 0; k < 256; k++) { h->ac_code[t][k] = tabs.ac[t].code[k]; h->ac_len[t][k] = tabs.ac[t].len[k]; }
    }
    HIP_TRY(d, hipMemcpyAsync((void *)a.tables, h, sizeof(*h), hipMemcpyHostToDevice, stream));
    return MIJPEG_OK;
  }

  // geometry, buffers, tables (optimised ones cost a synchronisation of their own), then count + prefix sums
  int stage_a(mijpeg_decoder *dec, const mijpeg_info &info, const int16_t *coef_dev, int ri, int optimize, int slot_, hipStream_t st)
  {
    d = dec; f = &info; slot = slot_; stream = st; restart_interval = ri;
    const int nc = info.components;
    memset(&a, 0, sizeof(a));
    a.coef = coef_dev;
    a.ncomp = nc;
    a.mcus_x = info.mcus_x;
    a.total_mcus = info.mcus_x * info.mcus_y;
    a.ri = ri ? ri : a.total_mcus;
    int B = 0;
    for (int c = 0; c < nc; c++) {
      a.hs[c] = nc > 1 ? info.hsamp[c] : 1;
      a.vs[c] = nc > 1 ? info.vsamp[c] : 1;
      a.bw[c] = info.blocks_w[c];
      a.nbx[c] = ((info.width + info.subx[c] - 1) / info.subx[c] + 7) >> 3;
      a.nby[c] = ((info.height + info.suby[c] - 1) / info.suby[c] + 7) >> 3;
      a.coef_off[c] = info.coef_offset[c];
      for (int by = 0; by < a.vs[c]; by++)
        for (int bx = 0; bx < a.hs[c]; bx++) {
          if (B >= 64) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "too many blocks per MCU for the device entropy coder");
          a.blk_comp[B] = (uint8_t)c;
          a.blk_bx[B] = (uint8_t)bx;
          a.blk_by[B] = (uint8_t)by;
          B++;
        }
    }
    a.blocks_per_mcu = B;
    const uint64_t nblocks = (uint64_t)a.total_mcus * (uint64_t)B;
    if (nblocks >= ((uint64_t)1 << 30)) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "frame too large for the device entropy coder");
    a.total_blocks = (uint32_t)nblocks;
    a.n_intervals = (uint32_t)((a.total_mcus + a.ri - 1) / a.ri);
    const uint32_t N = a.total_blocks, I = a.n_intervals;
    // arena 1: tables, statistics, block and interval arrays, scan scratch
    size_t o = 0;
    const size_t o_tab = o; o = al(o + sizeof(HencTables));
    const size_t o_hist = o; o = al(o + 4 * 256 * 4);
    const size_t o_bits = o; o = al(o + (size_t)N * 4);
    const size_t o_bitpos = o; o = al(o + ((size_t)N + 1) * 8);
    const size_t o_ibytes = o; o = al(o + (size_t)I * 4);
    const size_t o_istart = o; o = al(o + ((size_t)I + 1) * 8);
    const size_t scratch_words = ((size_t)N / 1024 + 8) * 2 + 8192;
    const size_t o_scratch = o; o = al(o + scratch_words * 8);
    int rc = ensure_dev(d, (void **)&d->henc_dev[slot], &d->henc_cap[slot], o);
    if (rc) return rc;
    if (!d->henc_host) HIP_TRY(d, hipHostMalloc((void **)&d->henc_host, 64 + 2 * sizeof(HencTables), hipHostMallocDefault));
    readback = d->henc_host + 2 * slot;
    uint8_t *base = d->henc_dev[slot];
    a.tables = (const HencTables *)(base + o_tab);
    a.hist = (uint32_t *)(base + o_hist);
    a.bits = (uint32_t *)(base + o_bits);
    a.bitpos = (const uint64_t *)(base + o_bitpos);
    a.ibytes = (uint32_t *)(base + o_ibytes);
    a.istart = (const uint64_t *)(base + o_istart);
    scratch = (uint64_t *)(base + o_scratch);
    enc_standard_tables(tabs);
    rc = upload_tables();
    if (rc) return rc;
    if (optimize) { // symbol statistics first, tables from them (Annex K.2)
      HIP_TRY(d, hipMemsetAsync(base + o_hist, 0, 4 * 256 * 4, stream));
      if (henc_count(a, true, stream)) return hip_fail(d, hipGetLastError(), "henc_count_kernel launch");
      uint32_t hist[4][256];
      HIP_TRY(d, hipMemcpyAsync(hist, base + o_hist, sizeof(hist), hipMemcpyDeviceToHost, stream));
      HIP_TRY(d, hipStreamSynchronize(stream));
      enc_optimal_tables(tabs, hist, hist + 2, nc > 1 ? 2 : 1);
      rc = upload_tables();
      if (rc) return rc;
    }
    if (henc_count(a, false, stream)) return hip_fail(d, hipGetLastError(), "henc_count_kernel launch");
    if (exclusive_scan_u32(a.bits, (uint64_t *)a.bitpos, N, scratch, stream)) return hip_fail(d, hipGetLastError(), "scan launch");
    if (henc_interval_bytes(a, stream)) return hip_fail(d, hipGetLastError(), "henc_interval_bytes_kernel launch");
    if (exclusive_scan_u32(a.ibytes, (uint64_t *)a.istart, I, scratch, stream)) return hip_fail(d, hipGetLastError(), "scan launch");
    HIP_TRY(d, hipMemcpyAsync(&readback[0], a.istart + I, 8, hipMemcpyDeviceToHost, stream));
    return MIJPEG_OK;
  }

  // plain stream, stuffing
  int stage_b()
  {
    HIP_TRY(d, hipStreamSynchronize(stream));
    const uint64_t plain_bytes = readback[0];
    const uint32_t I = a.n_intervals;
    // (coefficients the forward kernels make of 8-bit pixels always have a code: at most 11 / 10 bits, the host coder's check)
    chunks = (uint32_t)((plain_bytes + HENC_STUFF_CHUNK - 1) / HENC_STUFF_CHUNK);
    size_t q = 0;
    const size_t q_plain = q; q = al(q + (size_t)plain_bytes + 16);
    const size_t q_ffc = q; q = al(q + (size_t)chunks * 4 + 4);
    const size_t q_ffs = q; q = al(q + ((size_t)chunks + 1) * 8);
    const size_t q_out = q; q = al(q + (size_t)plain_bytes * 2 + (size_t)I * 2 + 16);
    const int rc = ensure_dev(d, (void **)&d->henc_out_dev[slot], &d->henc_out_cap[slot], q);
    if (rc) return rc;
    uint8_t *ob = d->henc_out_dev[slot];
    a.plain = (uint32_t *)(ob + q_plain);
    a.plain_bytes = plain_bytes;
    a.ffcount = (uint32_t *)(ob + q_ffc);
    a.ffstart = (const uint64_t *)(ob + q_ffs);
    a.out = ob + q_out;
    HIP_TRY(d, hipMemsetAsync(ob + q_plain, 0, al((size_t)plain_bytes + 16), stream));
    if (henc_emit(a, stream)) return hip_fail(d, hipGetLastError(), "henc_emit_kernel launch");
    if (henc_count_ff(a, stream)) return hip_fail(d, hipGetLastError(), "henc_count_ff_kernel launch");
    if (exclusive_scan_u32(a.ffcount, (uint64_t *)a.ffstart, chunks, scratch, stream)) return hip_fail(d, hipGetLastError(), "scan launch");
    if (henc_stuff(a, stream)) return hip_fail(d, hipGetLastError(), "henc_stuff_kernel launch");
    HIP_TRY(d, hipMemcpyAsync(&readback[1], a.ffstart + chunks, 8, hipMemcpyDeviceToHost, stream));
    return MIJPEG_OK;
  }

  // headers on the host, download of the entropy coded data behind them
  int stage_c()
  {
    HIP_TRY(d, hipStreamSynchronize(stream));
    ecs = (size_t)a.plain_bytes + (size_t)readback[1] + (size_t)(a.n_intervals - 1) * 2;
    head.clear();
    enc_write_headers(head, *f, tabs, restart_interval);
    head_size = head.size();
    result = (uint8_t *)malloc(head_size + ecs + 2);
    if (!result) return set_error(d, MIJPEG_ERR_OUT_OF_MEMORY, "out of memory for the stream");
    memcpy(result, head.data(), head_size);
    const hipError_t e = hipMemcpyAsync(result + head_size, a.out, ecs, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) { free(result); result = nullptr; return hip_fail(d, e, "download of the stream"); }
    return MIJPEG_OK;
  }

  int finish(uint8_t **out_stream, size_t *out_size)
  {
    const hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { free(result); result = nullptr; return hip_fail(d, e, "download of the stream"); }
    result[head_size + ecs] = 0xff;
    result[head_size + ecs + 1] = 0xd9;
    *out_stream = result;
    *out_size = head_size + ecs + 2;
    result = nullptr;
    return MIJPEG_OK;
  }
};

static int device_entropy_code(mijpeg_decoder *d, const mijpeg_info &f, const int16_t *coef_dev, int restart_interval, int optimize,
                               uint8_t **stream, size_t *size)
{
  HencJob job;
  int rc = job.stage_a(d, f, coef_dev, restart_interval, optimize, 0, d->stream);
  if (!rc) rc = job.stage_b();
  if (!rc) rc = job.stage_c();
  if (!rc) rc = job.finish(stream, size);
  return rc;
}

int mijpeg_encode_batch_device(mijpeg_decoder *d, const mijpeg_forward_batch *b, int restart_interval, int optimize, uint8_t **streams, size_t *sizes)
try {
  if (!d || !b || !streams || !sizes || b->frames < 1 || restart_interval < 0 || restart_interval > 65535) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "decoder was created without a device");
  HIP_TRY(d, hipSetDevice(d->device));
  for (int f = 0; f < b->frames; f++) { streams[f] = nullptr; sizes[f] = 0; }
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = mijpeg_launch_forward(b, d->stream);
  if (rc) return set_error(d, rc, "forward kernel launch failed");
  // frame f on stream f & 1 with buffer set f & 1: while the host waits for one frame's byte counts and download, the
  // other frame's kernels run
  if (!d->copy_stream) HIP_TRY(d, hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
  HIP_TRY(d, hipEventRecord(d->ev0, d->stream));
  HIP_TRY(d, hipStreamWaitEvent(d->copy_stream, d->ev0, 0));
  hipStream_t st[2] = {d->stream, d->copy_stream};
  HencJob jobs[2];
  auto coef_of = [&](int f) { return b->coef_dev + (int64_t)f * b->coef_frame_stride; };
  rc = jobs[0].stage_a(d, b->info, coef_of(0), restart_interval, optimize, 0, st[0]);
  for (int f = 0; f < b->frames && !rc; f++) {
    HencJob &cur = jobs[f & 1], &nxt = jobs[(f + 1) & 1];
    if (f + 1 < b->frames) rc = nxt.stage_a(d, b->info, coef_of(f + 1), restart_interval, optimize, (f + 1) & 1, st[(f + 1) & 1]);
    if (!rc) rc = cur.stage_b();
    if (!rc) rc = cur.stage_c();
    if (!rc) rc = cur.finish(&streams[f], &sizes[f]);
  }
  (void)hipStreamSynchronize(d->copy_stream);
  (void)hipStreamSynchronize(d->stream);
  if (rc)
    for (int f = 0; f < b->frames; f++) { free(streams[f]); streams[f] = nullptr; sizes[f] = 0; }
  d->timing[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); // mijpeg_last_timing: the whole call
  d->timing[1] = d->timing[2] = d->timing[3] = 0;
  return rc;
} catch (...) { return boundary_catch(d, "mijpeg_encode_batch_device"); }

int mijpeg_encode_image(mijpeg_decoder *d, const uint8_t *pixels, int32_t width, int32_t height, int32_t components, int64_t row_stride,
                        int quality, const int32_t *hsamp, const int32_t *vsamp, int restart_interval, int optimize, uint8_t **stream, size_t *size)
try {
  return mijpeg_encode_image_ex(d, pixels, width, height, components, row_stride, quality, hsamp, vsamp, restart_interval, optimize, 0, stream, size);
} catch (...) { return boundary_catch(d, "mijpeg_encode_image"); }

int mijpeg_encode_image_ex(mijpeg_decoder *d, const uint8_t *pixels, int32_t width, int32_t height, int32_t components, int64_t row_stride,
                           int quality, const int32_t *hsamp, const int32_t *vsamp, int restart_interval, int optimize, uint32_t flags,
                           uint8_t **stream, size_t *size)
try {
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  if (!d || !pixels || !stream || !size || (components != 1 && components != 3) || row_stride < (int64_t)width * components)
    return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0) return set_error(d, MIJPEG_ERR_NOT_AVAILABLE, "decoder was created without a device");
  HIP_TRY(d, hipSetDevice(d->device));
  mijpeg_forward_batch b;
  memset(&b, 0, sizeof(b));
  mijpeg_info &f = b.info;
  f.width = width;
  f.height = height;
  f.components = components;
  f.precision = 8;
  f.ycbcr = components == 3 ? 1 : 0;
  for (int c = 0; c < components; c++) {
    f.hsamp[c] = hsamp ? hsamp[c] : 1;
    f.vsamp[c] = vsamp ? vsamp[c] : 1;
    // the reference encoder defines a luma and a chroma table but its frame header selects table 0 for every component
    // (what its own files show: tests/test_encoder.py::test_quality_tables_are_the_reference_encoders), so that is
    // what reproduces its coefficients
    f.quant_index[c] = 0;
  }
  mijpeg_quality_tables(quality, f.quant[0], f.quant[1]);
  int rc = mijpeg_frame_layout(&f);
  if (rc) return set_error(d, rc, "invalid frame layout for encoding");
  const size_t px_bytes = (size_t)row_stride * (size_t)height, coef_bytes = (size_t)f.coef_count * sizeof(int16_t);
  rc = ensure_dev(d, (void **)&d->enc_dev, &d->enc_cap, px_bytes + 256 + coef_bytes);
  if (rc) return rc;
  int16_t *coef_dev = (int16_t *)(d->enc_dev + ((px_bytes + 255) & ~(size_t)255));
  // pinned staging: [pixels][coefficients].  The picture goes up in bands, each gathered into pinned memory by the pool
  // threads while the DMA of the previous band runs; the coefficients come down into pinned memory the coder reads.
  const size_t stage_bytes = ((px_bytes + 255) & ~(size_t)255) + coef_bytes;
  if (d->stage_cap < stage_bytes) {
    if (d->stage_host) (void)hipHostFree(d->stage_host);
    d->stage_host = nullptr;
    d->stage_cap = 0;
    HIP_TRY(d, hipHostMalloc((void **)&d->stage_host, stage_bytes, hipHostMallocDefault));
    d->stage_cap = stage_bytes;
  }
  int16_t *coef_host = (int16_t *)(d->stage_host + ((px_bytes + 255) & ~(size_t)255));
  {
    const size_t band = std::max<size_t>((size_t)8 << 20, (px_bytes + 7) / 8) & ~(size_t)255;
    for (size_t b0 = 0; b0 < px_bytes; b0 += band) {
      const size_t len = std::min(band, px_bytes - b0);
      const size_t pieces = (len + ((size_t)1 << 20) - 1) >> 20;
      const int workers = (int)std::min<size_t>(pieces, (size_t)std::min(default_threads(), 16));
      parallel_for(workers, [&](int w) {
        for (size_t k = (size_t)w; k < pieces; k += (size_t)workers) {
          const size_t o = b0 + (k << 20), n = std::min<size_t>((size_t)1 << 20, b0 + len - o);
          memcpy(d->stage_host + o, pixels + o, n);
        }
      });
      HIP_TRY(d, hipMemcpyAsync(d->enc_dev + b0, d->stage_host + b0, len, hipMemcpyHostToDevice, d->stream));
    }
  }
  b.pixels_dev = d->enc_dev;
  b.pixel_row_stride = row_stride;
  b.pixel_frame_stride = (int64_t)px_bytes;
  b.coef_dev = coef_dev;
  b.coef_frame_stride = f.coef_count;
  b.frames = 1;
  const auto t_up = clk::now(); // uploads enqueued (the gathering is synchronous)
  rc = mijpeg_launch_forward(&b, d->stream);
  if (rc) return set_error(d, rc, "forward kernel launch failed");
  static const bool env_host_coder = getenv("MIJPEG_ENTROPY_CODER") && !strcmp(getenv("MIJPEG_ENTROPY_CODER"), "host");
  if (!(flags & MIJPEG_ENCODE_HOST_CODER) && !env_host_coder) {
    if (restart_interval < 0 || restart_interval > 65535) return set_error(d, MIJPEG_ERR_INVALID_PARAMETER, "invalid restart interval");
    rc = device_entropy_code(d, f, coef_dev, restart_interval, optimize, stream, size);
    d->timing[0] = std::chrono::duration<double>(t_up - t_begin).count();
    d->timing[1] = std::chrono::duration<double>(clk::now() - t_up).count(); // kernels, entropy coder and download of the stream
    d->timing[2] = d->timing[3] = 0;
    if (rc != MIJPEG_ERR_NOT_AVAILABLE) return rc;
  }
  HIP_TRY(d, hipStreamSynchronize(d->stream));
  const auto t_kernel = clk::now();
  HIP_TRY(d, hipMemcpyAsync(coef_host, coef_dev, coef_bytes, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(d, hipStreamSynchronize(d->stream));
  const auto t_down = clk::now();
  rc = mijpeg_encode_coefficients(&f, coef_host, restart_interval, optimize, 0, stream, size);
  // mijpeg_last_timing: gather + upload, kernels (incl. the rest of the upload), download, entropy coder
  d->timing[0] = std::chrono::duration<double>(t_up - t_begin).count();
  d->timing[1] = std::chrono::duration<double>(t_kernel - t_up).count();
  d->timing[2] = std::chrono::duration<double>(t_down - t_kernel).count();
  d->timing[3] = std::chrono::duration<double>(clk::now() - t_down).count();
  if (rc) return set_error(d, rc, "entropy coding failed");
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_encode_image_ex"); }

// ------------------------------------------------------------------------------------------------
// decoder-object reconstruction
// ------------------------------------------------------------------------------------------------
static int ensure_dev(mijpeg_decoder *d, void **ptr, size_t *cap, size_t bytes)
{
  if (*cap >= bytes) return MIJPEG_OK;
  quiesce(d);
  release_big(d->device, false, *ptr, *cap);
  *ptr = nullptr;
  *cap = 0;
  size_t got = 0;
  if (void *p = buffer_cache().take(d->device, false, bytes, &got)) {
    *ptr = p;
    *cap = got;
    return MIJPEG_OK;
  }
  HIP_TRY(d, hipMalloc(ptr, bytes));
  *cap = bytes;
  return MIJPEG_OK;
}

// The frame as the reconstruction sees it: the whole picture, or -- without upsampling -- one component at its own
// resolution, which is a single-component identity-transformed frame over that component's coefficient plane
// (BlockBitmapRequester::ReconstructUnsampled with rr_bUpsampling = false: control/blockbitmaprequester.cpp:1013-1074,
// control/bitmapctrl.cpp:273-294; the colour transformation is off then, codestream/rectanglerequest.cpp:157-159).
static mijpeg_info view_of(const mijpeg_info &f, int comp)
{
  if (comp < 0) return f;
  mijpeg_info v = f;
  v.components = 1;
  v.width = (f.width + f.subx[comp] - 1) / f.subx[comp];
  v.height = (f.height + f.suby[comp] - 1) / f.suby[comp];
  v.hsamp[0] = v.vsamp[0] = v.subx[0] = v.suby[0] = 1;
  v.quant_index[0] = f.quant_index[comp];
  v.blocks_w[0] = f.blocks_w[comp];
  v.blocks_h[0] = f.blocks_h[comp];
  v.mcus_x = v.blocks_w[0];
  v.mcus_y = v.blocks_h[0];
  v.coef_offset[0] = 0;
  v.coef_count = (int64_t)v.blocks_w[0] * v.blocks_h[0] * 64;
  v.range_max[0] = f.range_max[comp];
  v.ycbcr = 0;
  return v;
}

static int reconstruct_view(mijpeg_decoder *d, int comp, void *dst_device, int64_t row_stride, uint32_t flags, int sync)
{
  HIP_TRY(d, hipSetDevice(d->device));
  mijpeg_batch b;
  memset(&b, 0, sizeof(b));
  b.info = view_of(d->host.info, comp);
  b.coef_dev = d->coef_dev + (comp < 0 ? 0 : d->host.info.coef_offset[comp]);
  b.coef_frame_stride = b.info.coef_count;
  b.out_dev = (uint8_t *)dst_device;
  b.out_row_stride = row_stride;
  b.out_frame_stride = row_stride * b.info.height;
  b.frames = 1;
  b.flags = flags;
  b.xt = d->host.is_xt() ? &d->host.xt : nullptr;
  const size_t ws = mijpeg_workspace_bytes(&b);
  if (ws) {
    int rc = ensure_dev(d, (void **)&d->ws_dev, &d->ws_cap, ws);
    if (rc) return rc;
    b.workspace = d->ws_dev;
    b.workspace_bytes = d->ws_cap;
  }
  const int rc = mijpeg_launch_reconstruct(&b, d->stream);
  if (rc) return set_error(d, rc, rc == MIJPEG_ERR_DEVICE ? std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError())
                                                           : std::string("reconstruction not available for this stream"));
  if (sync) HIP_TRY(d, hipStreamSynchronize(d->stream));
  return MIJPEG_OK;
}

int mijpeg_reconstruct_device(mijpeg_decoder *d, void *dst_device, int64_t row_stride, uint32_t flags, int sync)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0) return set_error(d, MIJPEG_ERR_DEVICE, "decoder was created without a device: no reconstruction path");
  if (!d->uploaded) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no decoded coefficients: call mijpeg_decode_coefficients first");
  if (!dst_device) return set_error(d, MIJPEG_ERR_INVALID_PARAMETER, "destination pointer is NULL");
  return reconstruct_view(d, -1, dst_device, row_stride, flags & ~(MIJPEG_FLAG_DEVICE_OUTPUT | MIJPEG_FLAG_NO_UPSAMPLING), sync);
} catch (...) { return boundary_catch(d, "mijpeg_reconstruct_device"); }

void *mijpeg_host_alloc(size_t bytes)
try {
  void *p = nullptr;
  return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
} catch (...) { (void)boundary_catch(nullptr, "mijpeg_host_alloc"); return nullptr; }

void mijpeg_host_free(void *p)
try {
  if (p) (void)hipHostFree(p);
} catch (...) { (void)boundary_catch(nullptr, "mijpeg_host_free"); }

int mijpeg_reconstruct_host(mijpeg_decoder *d, void *dst_host, int64_t row_stride, uint32_t flags)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0) return set_error(d, MIJPEG_ERR_DEVICE, "decoder was created without a device: no reconstruction path");
  if (!d->uploaded) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no decoded coefficients: call mijpeg_decode_coefficients first");
  if (!dst_host) return set_error(d, MIJPEG_ERR_INVALID_PARAMETER, "destination pointer is NULL");
  const mijpeg_info &f = d->host.info;
  const size_t line = (size_t)f.width * f.components * (f.sample_bytes > 0 ? f.sample_bytes : 1);
  if (row_stride < (int64_t)line) return set_error(d, MIJPEG_ERR_INVALID_PARAMETER, "row stride is smaller than a line of samples");
  HIP_TRY(d, hipSetDevice(d->device));
  const size_t row = (line + 7) & ~(size_t)7; // device image: 8-byte aligned lines -> wide stores in the kernel
  int rc = ensure_dev(d, (void **)&d->img_dev, &d->img_dev_cap, row * f.height);
  if (rc) return rc;
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  rc = mijpeg_reconstruct_device(d, d->img_dev, (int64_t)row, flags, 0);
  if (rc) return rc;
  if ((size_t)row_stride == row)
    HIP_TRY(d, hipMemcpyAsync(dst_host, d->img_dev, row * f.height, hipMemcpyDeviceToHost, d->stream));
  else
    HIP_TRY(d, hipMemcpy2DAsync(dst_host, (size_t)row_stride, d->img_dev, row, line, (size_t)f.height, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(d, hipStreamSynchronize(d->stream));
  d->timing[1] = 0;
  d->timing[2] = 0;
  d->timing[3] = std::chrono::duration<double>(clk::now() - t0).count(); // upload tail + kernel + D2H
  d->img_valid = false;
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_reconstruct_host"); }

static int serve_rect(mijpeg_decoder *d, int view, uint32_t flags, bool to_device, int32_t min_x, int32_t min_y, int32_t max_x, int32_t max_y,
                      int32_t min_comp, int32_t max_comp, void *const dst[MIJPEG_MAX_COMPONENTS],
                      const int32_t bytes_per_pixel[MIJPEG_MAX_COMPONENTS], const int32_t bytes_per_row[MIJPEG_MAX_COMPONENTS]);

int mijpeg_reconstruct_rect(mijpeg_decoder *d, int32_t min_x, int32_t min_y, int32_t max_x, int32_t max_y, int32_t min_comp,
                            int32_t max_comp, uint32_t flags, void *const dst[MIJPEG_MAX_COMPONENTS],
                            const int32_t bytes_per_pixel[MIJPEG_MAX_COMPONENTS],
                            const int32_t bytes_per_row[MIJPEG_MAX_COMPONENTS])
try {
  if (!d || !dst || !bytes_per_pixel || !bytes_per_row) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0) return set_error(d, MIJPEG_ERR_DEVICE, "decoder was created without a device: no reconstruction path");
  if (!d->uploaded) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no decoded coefficients: call mijpeg_decode_coefficients first");
  const bool to_device = (flags & MIJPEG_FLAG_DEVICE_OUTPUT) != 0;
  const bool unsampled = (flags & MIJPEG_FLAG_NO_UPSAMPLING) != 0;
  flags &= ~(MIJPEG_FLAG_DEVICE_OUTPUT | MIJPEG_FLAG_NO_UPSAMPLING);
  int view = -1; // component whose own sample grid is reconstructed, -1: the upsampled picture
  if (unsampled) {
    // control/bitmapctrl.cpp:273-294: one component at a time, no colour transformation, and the rectangle (given
    // on the canvas) shrinks to the component's grid
    if (min_comp < 0) min_comp = 0;
    if (max_comp >= d->host.info.components) max_comp = d->host.info.components - 1;
    if (min_comp != max_comp)
      return set_error(d, MIJPEG_ERR_INVALID_PARAMETER, "if upsampling is disabled, components can only be reconstructed one by one");
    if (d->host.is_xt())
      return set_error(d, MIJPEG_ERR_OPERATION_UNIMPLEMENTED, "JPEG XT frames are not reconstructed without upsampling (the reference merges the component with residual scratch buffers it never initialised)");
    view = min_comp;
    flags |= MIJPEG_FLAG_NO_COLOR_TRANSFORM;
    const int sx = d->host.info.subx[view], sy = d->host.info.suby[view];
    min_x = (std::max(min_x, 0) + sx - 1) / sx;
    max_x = (max_x + sx) / sx - 1;
    min_y = (std::max(min_y, 0) + sy - 1) / sy;
    max_y = (max_y + sy) / sy - 1;
  }
  void *vdst[MIJPEG_MAX_COMPONENTS] = {dst[0], dst[1], dst[2], dst[3]};
  int32_t vbpp[MIJPEG_MAX_COMPONENTS] = {bytes_per_pixel[0], bytes_per_pixel[1], bytes_per_pixel[2], bytes_per_pixel[3]};
  int32_t vbpr[MIJPEG_MAX_COMPONENTS] = {bytes_per_row[0], bytes_per_row[1], bytes_per_row[2], bytes_per_row[3]};
  if (view >= 0) { // the view has one component, number 0
    vdst[0] = dst[view];
    vbpp[0] = bytes_per_pixel[view];
    vbpr[0] = bytes_per_row[view];
    min_comp = max_comp = 0;
  }
  return serve_rect(d, view, flags, to_device, min_x, min_y, max_x, max_y, min_comp, max_comp, vdst, vbpp, vbpr);
} catch (...) { return boundary_catch(d, "mijpeg_reconstruct_rect"); }

// The rectangle [min_x, max_x] x [min_y, max_y] (on the grid of `view`: the canvas, or a component's own samples) of the
// plain picture, components [min_comp, max_comp] of the view, into the bitmaps.
static int serve_rect(mijpeg_decoder *d, int view, uint32_t flags, bool to_device, int32_t min_x, int32_t min_y, int32_t max_x, int32_t max_y,
                      int32_t min_comp, int32_t max_comp, void *const dst[MIJPEG_MAX_COMPONENTS],
                      const int32_t bytes_per_pixel[MIJPEG_MAX_COMPONENTS], const int32_t bytes_per_row[MIJPEG_MAX_COMPONENTS])
{
  const mijpeg_info f = view_of(d->host.info, view);
  const int sb = f.sample_bytes > 0 ? f.sample_bytes : 1; // bytes per sample
  const int nc = f.components;
  // the whole frame is reconstructed once per (stream, flags, view) and then served rectangle by rectangle,
  // which is what the stripe loop of cmd/reconstruct.cpp:334-342 asks for
  const size_t row = ((size_t)f.width * nc * sb + 7) & ~(size_t)7;
  const size_t padded = row * f.height;
  using clk = std::chrono::steady_clock;
  if (!d->img_valid || d->img_flags != flags || d->img_view != view) {
    HIP_TRY(d, hipSetDevice(d->device));
    int rc = ensure_dev(d, (void **)&d->img_dev, &d->img_dev_cap, padded);
    if (rc) return rc;
    auto t0 = clk::now();
    HIP_TRY(d, hipStreamSynchronize(d->stream)); // uploads complete
    auto t1 = clk::now();
    rc = reconstruct_view(d, view, d->img_dev, (int64_t)row, flags, to_device ? 1 : 0); // host requests: the copy below follows in stream order
    if (rc) return rc;
    d->timing[1] = std::chrono::duration<double>(t1 - t0).count();
    d->timing[2] = std::chrono::duration<double>(clk::now() - t1).count();
    d->timing[3] = 0;
    d->img_valid = true;
    d->img_host_valid = false;
    d->img_flags = flags;
    d->img_view = view;
  }
  if (!to_device && !d->img_host_valid) {
    HIP_TRY(d, hipSetDevice(d->device));
    if (d->img_host_cap < padded) {
      quiesce(d);
      release_big(d->device, true, d->img_host, d->img_host_cap);
      d->img_host = nullptr;
      d->img_host_cap = 0;
      size_t got = 0;
      if (void *p = buffer_cache().take(d->device, true, padded, &got)) {
        d->img_host = (uint8_t *)p;
        d->img_host_cap = got;
      } else {
        HIP_TRY(d, hipHostMalloc((void **)&d->img_host, padded, hipHostMallocDefault));
        d->img_host_cap = padded;
      }
    }
    // bands of about 4 MiB (at least 8 lines): enqueue all of them now, wait for them as they are asked for
    static const long band_mib = getenv("MIJPEG_RECT_BAND_MIB") ? atol(getenv("MIJPEG_RECT_BAND_MIB")) : 4; // tuning; <= 0: one band
    d->band_lines = band_mib <= 0 ? f.height : (int)std::max<size_t>(8, (((size_t)band_mib << 20) / std::max<size_t>(row, 1) + 7) & ~(size_t)7);
    d->bands = (f.height + d->band_lines - 1) / d->band_lines;
    while ((int)d->band_events.size() < d->bands) {
      hipEvent_t e;
      HIP_TRY(d, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      d->band_events.push_back(e);
    }
    for (int b = 0; b < d->bands; b++) {
      const size_t y0 = (size_t)b * d->band_lines, y1 = std::min<size_t>(f.height, y0 + d->band_lines);
      HIP_TRY(d, hipMemcpyAsync(d->img_host + y0 * row, d->img_dev + y0 * row, (y1 - y0) * row, hipMemcpyDeviceToHost, d->stream));
      HIP_TRY(d, hipEventRecord(d->band_events[(size_t)b], d->stream));
    }
    d->bands_waited = 0;
    d->img_host_valid = true;
  }
  // bands [0, upto] of the host copy have arrived when this returns
  auto wait_bands = [&](int upto) -> int {
    auto t2 = clk::now();
    for (; d->bands_waited <= upto && d->bands_waited < d->bands; d->bands_waited++)
      HIP_TRY(d, hipEventSynchronize(d->band_events[(size_t)d->bands_waited]));
    d->timing[3] += std::chrono::duration<double>(clk::now() - t2).count();
    return MIJPEG_OK;
  };
  if (min_x < 0) min_x = 0;
  if (min_y < 0) min_y = 0;
  if (max_x >= f.width) max_x = f.width - 1;
  if (max_y >= f.height) max_y = f.height - 1;
  if (min_comp < 0) min_comp = 0;
  if (max_comp >= nc) max_comp = nc - 1;
  // an empty request after clipping is served by doing nothing (codestream/rectanglerequest.cpp clips alike)
  if (min_x > max_x || min_y > max_y || min_comp > max_comp) return MIJPEG_OK;
  if (to_device) {
    ScatterArgs a;
    memset(&a, 0, sizeof(a));
    a.src = d->img_dev;
    a.src_row = (int64_t)row;
    a.ncomp = nc;
    a.sample_bytes = sb;
    a.x0 = min_x;
    a.y0 = min_y;
    a.w = max_x - min_x + 1;
    a.h = max_y - min_y + 1;
    a.c0 = min_comp;
    a.c1 = max_comp;
    for (int c = 0; c < nc; c++) {
      a.dst[c] = (uint8_t *)dst[c];
      a.bytes_per_pixel[c] = bytes_per_pixel[c];
      a.bytes_per_row[c] = bytes_per_row[c];
    }
    HIP_TRY(d, hipSetDevice(d->device));
    if (launch_scatter_rect(a, d->stream)) return hip_fail(d, hipGetLastError(), "scatter_rect_kernel launch");
    HIP_TRY(d, hipStreamSynchronize(d->stream));
    return MIJPEG_OK;
  }
  // interleaved destination (the layout cmd/bitmaphook.cpp hands out): whole lines at once
  bool interleaved = min_comp == 0 && max_comp == nc - 1 && dst[0];
  for (int c = 0; c < nc && interleaved; c++)
    interleaved = dst[c] == (uint8_t *)dst[0] + c * sb && bytes_per_pixel[c] == nc * sb && bytes_per_row[c] == bytes_per_row[0];
  if (interleaved) {
    const size_t line = (size_t)(max_x - min_x + 1) * nc * sb;
    const int lines = max_y - min_y + 1;
    auto copy_lines = [&](int y0, int y1) {
      for (int y = y0; y < y1; y++)
        memcpy((uint8_t *)dst[0] + (ptrdiff_t)y * bytes_per_row[0] + (ptrdiff_t)min_x * nc * sb,
               d->img_host + (size_t)y * row + (size_t)min_x * nc * sb, line);
    };
    // big rectangles (whole frames) are copied by the worker pool, one memcpy stream per worker, in a few slabs: the
    // workers copy slab k while the bands of slab k + 1 are still arriving
    const int parts = (int)std::min<size_t>((size_t)std::min(default_threads(), 16), line * lines / (4u << 20));
    if (parts > 1) {
      const int slabs = std::min(4, std::max(1, lines / (8 * d->band_lines)));
      for (int k = 0; k < slabs; k++) {
        const int s0 = min_y + (int)((int64_t)lines * k / slabs), s1 = min_y + (int)((int64_t)lines * (k + 1) / slabs);
        if (const int rc = wait_bands((s1 - 1) / d->band_lines)) return rc;
        parallel_for(parts, [&](int i) { copy_lines(s0 + (int)((int64_t)(s1 - s0) * i / parts), s0 + (int)((int64_t)(s1 - s0) * (i + 1) / parts)); });
      }
    } else {
      if (const int rc = wait_bands(max_y / d->band_lines)) return rc;
      copy_lines(min_y, max_y + 1);
    }
    return MIJPEG_OK;
  }
  if (const int rc = wait_bands(max_y / d->band_lines)) return rc;
  for (int c = min_comp; c <= max_comp; c++) {
    if (!dst[c]) continue;
    for (int y = min_y; y <= max_y; y++) {
      const uint8_t *src = d->img_host + (size_t)y * row + ((size_t)min_x * nc + c) * sb;
      uint8_t *out = (uint8_t *)dst[c] + (ptrdiff_t)y * bytes_per_row[c] + (ptrdiff_t)min_x * bytes_per_pixel[c];
      const int n = max_x - min_x + 1;
      const int bpp = bytes_per_pixel[c];
      if (sb == 1) {
        if (nc == 1 && bpp == 1) memcpy(out, src, (size_t)n);
        else
          for (int x = 0; x < n; x++) out[(ptrdiff_t)x * bpp] = src[(size_t)x * nc];
      } else {
        for (int x = 0; x < n; x++) memcpy(out + (ptrdiff_t)x * bpp, src + (size_t)x * nc * 2, 2);
      }
    }
  }
  return MIJPEG_OK;
}

// ------------------------------------------------------------------------------------------------
// JPEG::DisplayRectangle as a sequence of calls: request_model.hpp plans, this serves
// ------------------------------------------------------------------------------------------------
// Frame description for a request that does not show the plain picture: the whole picture, or -- without upsampling --
// the grid of component `view` with every component of the frame on it (the ones that were not asked for are zeros; only
// a colour transformer left over from earlier upsampled requests makes them matter).
static mijpeg_info request_frame(const mijpeg_info &f, int view, bool all_components)
{
  if (view < 0) return f;
  if (!all_components) return view_of(f, view);
  mijpeg_info v = f;
  v.width = (f.width + f.subx[view] - 1) / f.subx[view];
  v.height = (f.height + f.suby[view] - 1) / f.suby[view];
  for (int c = 0; c < f.components; c++) {
    v.hsamp[c] = v.vsamp[c] = v.subx[c] = v.suby[c] = 1;
    v.blocks_w[c] = f.blocks_w[view];
    v.blocks_h[c] = f.blocks_h[view];
    v.coef_offset[c] = f.coef_offset[view]; // read only where the row map says so: component `view`
    v.quant_index[c] = f.quant_index[view];
    v.range_max[c] = f.range_max[view];
  }
  v.mcus_x = v.blocks_w[0];
  v.mcus_y = v.blocks_h[0];
  return v;
}

// The request models of a decoded image start with its first DisplayRectangle call (every decode resets them): one for a plain
// frame; two for a JPEG XT frame -- legacy and residual image share m_bSubsampling (request_model.hpp)
static void ensure_request_models(mijpeg_decoder *d)
{
  if (d->model_valid) return;
  const mijpeg_info &f = d->host.info;
  if (d->host.is_xt() && f.components == 3) {
    const mijpeg_info &r = d->host.xt.residual;
    bool lsub = false, rsub = false;
    for (int c = 0; c < 3; c++) {
      lsub = lsub || f.subx[c] > 1 || f.suby[c] > 1;
      rsub = rsub || r.subx[c] > 1 || r.suby[c] > 1;
    }
    d->model.reset(3, f.width, f.height, f.subx, f.suby, true, false, nullptr, nullptr, rsub);
    d->rmodel.reset(3, f.width, f.height, r.subx, r.suby, true, false, nullptr, nullptr, lsub);
  } else
    d->model.reset(f.components, f.width, f.height, f.subx, f.suby, f.ycbcr != 0, f.dnl != 0, f.rows, f.blocks_h);
  d->model_valid = true;
}

// The lines a request reconstructed into d->req_dev (row bytes each, nc interleaved samples of sb bytes) go out to the client's
// bitmaps: columns [min_x, cx1[c]], lines [min_y, cy1[c]] of component c
static int hand_out_request(mijpeg_decoder *d, int min_x, int min_y, int y_count, const int32_t *cx1, const int32_t *cy1, int min_comp, int max_comp,
                            int nc, int sb, size_t row, size_t padded, int vc, bool all_on_view, bool to_device, void *const *dst, const int32_t *bpp,
                            const int32_t *bpr)
{
  // hand the lines out
  for (int c = min_comp; c <= max_comp; c++) {
    if (!dst[c] || cx1[c] < min_x || cy1[c] < min_y) continue;
    const int plane = vc >= 0 ? (all_on_view ? c : 0) : c;
    if (to_device) {
      ScatterArgs a;
      memset(&a, 0, sizeof(a));
      a.src = d->req_dev;
      a.src_row = (int64_t)row;
      a.ncomp = nc;
      a.sample_bytes = sb;
      a.x0 = min_x;
      a.y0 = min_y;
      a.w = cx1[c] - min_x + 1;
      a.h = cy1[c] - min_y + 1;
      a.c0 = a.c1 = plane;
      a.dst[plane] = (uint8_t *)dst[c];
      a.bytes_per_pixel[plane] = bpp[c];
      a.bytes_per_row[plane] = bpr[c];
      if (launch_scatter_rect(a, d->stream)) return hip_fail(d, hipGetLastError(), "scatter_rect_kernel launch");
    }
  }
  if (to_device) {
    HIP_TRY(d, hipStreamSynchronize(d->stream));
    return MIJPEG_OK;
  }
  if (d->req_host_cap < padded) {
    if (d->req_host) (void)hipHostFree(d->req_host);
    d->req_host = nullptr;
    d->req_host_cap = 0;
    HIP_TRY(d, hipHostMalloc((void **)&d->req_host, padded, hipHostMallocDefault));
    d->req_host_cap = padded;
  }
  HIP_TRY(d, hipMemcpyAsync(d->req_host + (size_t)min_y * row, d->req_dev + (size_t)min_y * row, (size_t)y_count * row, hipMemcpyDeviceToHost,
                            d->stream));
  HIP_TRY(d, hipStreamSynchronize(d->stream));
  for (int c = min_comp; c <= max_comp; c++) {
    if (!dst[c] || cx1[c] < min_x || cy1[c] < min_y) continue;
    const int plane = vc >= 0 ? (all_on_view ? c : 0) : c;
    const int n = cx1[c] - min_x + 1;
    for (int y = min_y; y <= cy1[c]; y++) {
      const uint8_t *src = d->req_host + (size_t)y * row + ((size_t)min_x * nc + plane) * sb;
      uint8_t *out = (uint8_t *)dst[c] + (ptrdiff_t)y * bpr[c] + (ptrdiff_t)min_x * bpp[c];
      if (sb == 1)
        for (int x = 0; x < n; x++) out[(ptrdiff_t)x * bpp[c]] = src[(size_t)x * nc];
      else
        for (int x = 0; x < n; x++) memcpy(out + (ptrdiff_t)x * bpp[c], src + (size_t)x * nc * 2, 2);
    }
  }
  return MIJPEG_OK;
}

int mijpeg_display_rect(mijpeg_decoder *d, int32_t min_x, int32_t min_y, int32_t max_x, int32_t max_y, int32_t min_comp, int32_t max_comp,
                        uint32_t flags, const mijpeg_bitmap bitmaps[MIJPEG_MAX_COMPONENTS])
try {
  if (!d || !bitmaps) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->device < 0) return set_error(d, MIJPEG_ERR_DEVICE, "decoder was created without a device: no reconstruction path");
  if (!d->uploaded) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no decoded coefficients: call mijpeg_decode_coefficients first");
  const mijpeg_info &f = d->host.info;
  const bool to_device = (flags & MIJPEG_FLAG_DEVICE_OUTPUT) != 0;
  const bool upsample = !(flags & MIJPEG_FLAG_NO_UPSAMPLING);
  const bool ctrafo = !(flags & MIJPEG_FLAG_NO_COLOR_TRANSFORM);
  const uint32_t pass = flags & (MIJPEG_FLAG_FORCE_GENERIC | MIJPEG_FLAG_FORCE_SAFE);
  if (min_comp < 0) min_comp = 0;
  if (max_comp >= f.components) max_comp = f.components - 1;
  if (!upsample && min_comp != max_comp && min_comp <= max_comp)
    return set_error(d, MIJPEG_ERR_INVALID_PARAMETER, "if upsampling is disabled, components can only be reconstructed one by one");
  void *dst[MIJPEG_MAX_COMPONENTS];
  int32_t bpp[MIJPEG_MAX_COMPONENTS], bpr[MIJPEG_MAX_COMPONENTS];
  uint32_t bm_h[MIJPEG_MAX_COMPONENTS], bm_w[MIJPEG_MAX_COMPONENTS];
  for (int c = 0; c < MIJPEG_MAX_COMPONENTS; c++) {
    dst[c] = bitmaps[c].data;
    bpp[c] = bitmaps[c].bytes_per_pixel;
    bpr[c] = bitmaps[c].bytes_per_row;
    bm_w[c] = bitmaps[c].width;
    bm_h[c] = bitmaps[c].height;
  }
  if (d->host.is_xt()) {
    // JPEG XT: the residual image has row cursors and upsamplers of its own beside the legacy image's
    // (control/blockbitmaprequester.cpp:228-232, 356-372, 1118-1146, 1197-1222): one request model per image, fed the same
    // requests.  In the contract: what the reference's command line asks for -- all three components, upsampling and colour
    // transformation on -- with any order and size of rectangles.  (A component subset merges with whatever m_ppDTemp holds from
    // the block before, a request without the transformation builds another transformer: served as the plain picture.)
    const mijpeg_xt_params &x = d->host.xt;
    auto plain_picture = [&]() -> int {
      uint32_t maxmcu = 0xffffffffu;
      for (int c = min_comp; c <= max_comp; c++) maxmcu = std::min(maxmcu, (bm_h[c] >> 3) - 1u);
      if (maxmcu != 0xffffffffu && (int64_t)max_y > (int64_t)maxmcu * 8 + 7) max_y = (int32_t)(maxmcu * 8 + 7);
      if (max_y < min_y) return MIJPEG_OK;
      return mijpeg_reconstruct_rect(d, min_x, min_y, max_x, max_y, min_comp, max_comp, flags, dst, bpp, bpr);
    };
    if (!upsample || !ctrafo || min_comp != 0 || max_comp != 2 || f.components != 3) return plain_picture();
    const mijpeg_info &r = x.residual;
    ensure_request_models(d);
    const RequestPlan pl = d->model.request(min_x, min_y, max_x, max_y, 0, 2, true, true, bm_h);
    const RequestPlan pr = x.no_residual ? pl : d->rmodel.request(min_x, min_y, max_x, max_y, 0, 2, true, true, bm_h);
    if (pl.nothing) return MIJPEG_OK;
    // a residual component without an upsampler whose cursor stands behind its last row: `rrow->BlockAt(x)` on a NULL row
    // (:1057-1058, :1201-1202) -- the reference does not survive this request
    if (!x.no_residual)
      for (int c = 0; c < 3; c++)
        if (!(pr.upsampling_path && pr.upsampler[c]))
          for (int g = pr.g0[c]; g <= pr.g1[c]; g++)
            if (g >= (int)pr.rowmap[c].size() || pr.rowmap[c][(size_t)g] < 0)
              return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST,
                               "the request walks the residual image's row cursor behind its last row (the reference dereferences a NULL row here)");
    // BitmapCtrl::ExtractBitmap (interface/imagebitmap.cpp:58-129): blocks whose corner lies outside the bitmap the hook described
    // are not written
    int32_t cx1[MIJPEG_MAX_COMPONENTS], cy1[MIJPEG_MAX_COMPONENTS];
    for (int c = 0; c < 3; c++) {
      auto last_in = [](int32_t lo, int32_t hi, uint32_t extent) -> int32_t {
        if ((uint32_t)lo >= extent) return lo - 1;
        const int64_t last_block = ((int64_t)extent - 1) >> 3;
        return (int32_t)std::min<int64_t>(hi, std::max<int64_t>(last_block, lo >> 3) * 8 + 7);
      };
      cx1[c] = last_in(pl.min_x, pl.max_x, bm_w[c]);
      cy1[c] = last_in(pl.min_y, pl.max_y, bm_h[c]);
    }
    if (pl.plain && (x.no_residual || pr.plain)) {
      for (int c = 0; c <= 2;) { // components with the same writable extent go out together (all of them, normally)
        int e = c;
        while (e + 1 <= 2 && cx1[e + 1] == cx1[c] && cy1[e + 1] == cy1[c]) e++;
        if (cx1[c] >= pl.min_x && cy1[c] >= pl.min_y) {
          const int rc = mijpeg_reconstruct_rect(d, pl.min_x, pl.min_y, cx1[c], cy1[c], c, e, flags, dst, bpp, bpr);
          if (rc) return rc;
        }
        c = e + 1;
      }
      return MIJPEG_OK;
    }
    // ---- not the plain picture: both images through the unfused kernels with their row maps on this request's lines
    HIP_TRY(d, hipSetDevice(d->device));
    mijpeg_batch b;
    memset(&b, 0, sizeof(b));
    b.info = f;
    b.xt = &x;
    const int sb = f.sample_bytes > 0 ? f.sample_bytes : 2;
    const size_t row = ((size_t)f.width * 3 * sb + 7) & ~(size_t)7, padded = row * f.height;
    int rc = ensure_dev(d, (void **)&d->req_dev, &d->req_dev_cap, padded);
    if (rc) return rc;
    int stride = 1;
    for (int c = 0; c < 3; c++) stride = std::max(stride, std::max(f.blocks_h[c], r.blocks_h[c]));
    std::vector<int32_t> maps((size_t)6 * stride);
    for (int pn = 0; pn < 6; pn++) {
      const RequestPlan &p = pn < 3 ? pl : pr;
      const int c = pn % 3;
      int32_t *m = maps.data() + (size_t)pn * stride;
      for (int g = 0; g < stride; g++) m[g] = g;
      if (pn >= 3 && x.no_residual) continue;
      for (int g = p.g0[c]; g <= p.g1[c] && g < stride && g < (int)p.rowmap[c].size(); g++) m[g] = p.rowmap[c][(size_t)g];
    }
    rc = ensure_dev(d, (void **)&d->rowmap_dev, &d->rowmap_cap, maps.size() * sizeof(int32_t));
    if (rc) return rc;
    HIP_TRY(d, hipMemcpyAsync(d->rowmap_dev, maps.data(), maps.size() * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
    HIP_TRY(d, hipStreamSynchronize(d->stream)); // `maps` is pageable and leaves scope
    b.coef_dev = d->coef_dev;
    b.coef_frame_stride = f.coef_count;
    b.out_dev = d->req_dev;
    b.out_row_stride = (int64_t)row;
    b.out_frame_stride = (int64_t)padded;
    b.frames = 1;
    b.flags = pass | MIJPEG_FLAG_FORCE_GENERIC;
    const size_t ws = mijpeg_workspace_bytes(&b);
    if (ws) {
      rc = ensure_dev(d, (void **)&d->ws_dev, &d->ws_cap, ws);
      if (rc) return rc;
      b.workspace = d->ws_dev;
      b.workspace_bytes = d->ws_cap;
    }
    const int y_count = pl.max_y - pl.min_y + 1;
    RequestExtra rx;
    memset(&rx, 0, sizeof(rx));
    rx.rowmap_dev = d->rowmap_dev;
    rx.rowmap_stride = stride;
    rx.corner_x = pl.corner_x;
    rx.corner_y = pl.corner_y;
    rx.y_base = pl.min_y;
    rx.y_count = y_count;
    rx.ycc = 1;
    for (int pn = 0; pn < 6; pn++) {
      const RequestPlan &p = pn < 3 ? pl : pr;
      const mijpeg_info &g = pn < 3 ? f : r;
      const int c = pn % 3;
      const bool up = p.upsampling_path && p.upsampler[c] && !(pn >= 3 && x.no_residual);
      rx.wstart[pn] = up ? p.wstart[c] : 0;
      rx.wlimit[pn] = up ? p.wlimit[c] : (f.height + g.suby[c] - 1) / std::max(1, g.suby[c]);
    }
    rc = launch_reconstruct_ex(&b, d->stream, &rx);
    if (rc) return set_error(d, rc, rc == MIJPEG_ERR_DEVICE ? std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError())
                                                             : std::string("reconstruction not available for this request"));
    return hand_out_request(d, pl.min_x, pl.min_y, y_count, cx1, cy1, 0, 2, 3, sb, row, padded, -1, false, to_device, dst, bpp, bpr);
  }
  ensure_request_models(d);
  const RequestPlan p = d->model.request(min_x, min_y, max_x, max_y, min_comp, max_comp, upsample, ctrafo, bm_h);
  if (p.nothing) return MIJPEG_OK;
  // BitmapCtrl::ExtractBitmap (interface/imagebitmap.cpp:58-129): a block whose corner lies outside the bitmap the hook
  // described is blank -- nothing of it is written; one that starts inside is written in full
  int32_t cx1[MIJPEG_MAX_COMPONENTS], cy1[MIJPEG_MAX_COMPONENTS];
  auto last_inside = [](int32_t lo, int32_t hi, uint32_t extent) -> int32_t { // last sample of the blocks whose corner is below extent
    if ((uint32_t)lo >= extent) return lo - 1;
    const int64_t last_block = ((int64_t)extent - 1) >> 3; // the last block whose (aligned) corner is inside
    return (int32_t)std::min<int64_t>(hi, std::max<int64_t>(last_block, lo >> 3) * 8 + 7);
  };
  for (int c = 0; c < f.components; c++) {
    cx1[c] = last_inside(p.min_x, p.max_x, bm_w[c]);
    cy1[c] = last_inside(p.min_y, p.max_y, bm_h[c]);
  }
  const int vc = p.view; // without upsampling the single component of the view is number 0 there
  if (p.plain) {
    const uint32_t fl = pass | (p.ycc || !f.ycbcr ? 0u : MIJPEG_FLAG_NO_COLOR_TRANSFORM) | (vc >= 0 ? MIJPEG_FLAG_NO_COLOR_TRANSFORM : 0u);
    // components with the same writable extent go out together (all of them, normally: one interleaved copy)
    for (int c = min_comp; c <= max_comp;) {
      int e = c;
      while (e + 1 <= max_comp && cx1[e + 1] == cx1[c] && cy1[e + 1] == cy1[c]) e++;
      if (cx1[c] >= p.min_x && cy1[c] >= p.min_y) {
        int rc;
        if (vc >= 0) {
          void *vdst[MIJPEG_MAX_COMPONENTS] = {dst[vc], nullptr, nullptr, nullptr};
          int32_t vbpp[MIJPEG_MAX_COMPONENTS] = {bpp[vc], 0, 0, 0}, vbpr[MIJPEG_MAX_COMPONENTS] = {bpr[vc], 0, 0, 0};
          rc = serve_rect(d, vc, fl, to_device, p.min_x, p.min_y, cx1[c], cy1[c], 0, 0, vdst, vbpp, vbpr);
        } else
          rc = serve_rect(d, -1, fl, to_device, p.min_x, p.min_y, cx1[c], cy1[c], c, e, dst, bpp, bpr);
        if (rc) return rc;
      }
      c = e + 1;
    }
    return MIJPEG_OK;
  }
  // ---- not the plain picture: row maps, zeros, displaced upsampler output -> the generic kernels on this request's lines
  HIP_TRY(d, hipSetDevice(d->device));
  const bool all_on_view = vc >= 0 && p.ycc;
  mijpeg_batch b;
  memset(&b, 0, sizeof(b));
  b.info = request_frame(f, vc, all_on_view);
  b.info.ycbcr = p.ycc ? 1 : 0;
  const mijpeg_info &g = b.info;
  const int nc = g.components, sb = g.sample_bytes > 0 ? g.sample_bytes : 1;
  const size_t row = ((size_t)g.width * nc * sb + 7) & ~(size_t)7, padded = row * g.height;
  int rc = ensure_dev(d, (void **)&d->req_dev, &d->req_dev_cap, padded);
  if (rc) return rc;
  // row maps: identity outside what the plan defines; components that were not asked for are zeros
  int stride = 1;
  for (int c = 0; c < nc; c++) stride = std::max(stride, g.blocks_h[c]);
  std::vector<int32_t> maps((size_t)nc * stride);
  bool all_zero = !p.ycc;
  for (int c = 0; c < nc; c++) {
    const int pc = vc >= 0 ? (all_on_view ? c : vc) : c; // component of the frame behind plane c of the request frame
    int32_t *m = maps.data() + (size_t)c * stride;
    for (int r = 0; r < stride; r++) m[r] = r;
    if (!p.requested[pc]) {
      for (int r = 0; r < stride; r++) m[r] = -1;
      continue;
    }
    // (a component that appears in no scan carries coefficients that transform to zeros in every row: host_decoder.cpp)
    for (int r = p.g0[pc]; r <= p.g1[pc] && r < stride && r < (int)p.rowmap[pc].size(); r++) {
      m[r] = p.rowmap[pc][(size_t)r];
      if (m[r] >= 0) all_zero = false;
    }
  }
  const int y_count = p.max_y - p.min_y + 1;
  if (all_zero) {
    // every sample the request shows is the transform of "no row": 0 through the filters and the identity transformation
    HIP_TRY(d, hipMemsetAsync(d->req_dev + (size_t)p.min_y * row, 0, (size_t)y_count * row, d->stream));
  } else {
    rc = ensure_dev(d, (void **)&d->rowmap_dev, &d->rowmap_cap, maps.size() * sizeof(int32_t));
    if (rc) return rc;
    HIP_TRY(d, hipMemcpyAsync(d->rowmap_dev, maps.data(), maps.size() * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
    HIP_TRY(d, hipStreamSynchronize(d->stream)); // `maps` is pageable and leaves scope; uploads of the coefficients are complete too
    b.coef_dev = d->coef_dev + (vc >= 0 && !all_on_view ? f.coef_offset[vc] : 0);
    b.coef_frame_stride = g.coef_count;
    b.out_dev = d->req_dev;
    b.out_row_stride = (int64_t)row;
    b.out_frame_stride = (int64_t)padded;
    b.frames = 1;
    b.flags = pass | MIJPEG_FLAG_FORCE_GENERIC | (p.ycc ? 0u : MIJPEG_FLAG_NO_COLOR_TRANSFORM);
    const size_t ws = mijpeg_workspace_bytes(&b);
    if (ws) {
      rc = ensure_dev(d, (void **)&d->ws_dev, &d->ws_cap, ws);
      if (rc) return rc;
      b.workspace = d->ws_dev;
      b.workspace_bytes = d->ws_cap;
    }
    RequestExtra rx;
    memset(&rx, 0, sizeof(rx));
    rx.rowmap_dev = d->rowmap_dev;
    rx.rowmap_stride = stride;
    rx.corner_x = p.corner_x;
    rx.corner_y = p.corner_y;
    rx.y_base = p.min_y;
    rx.y_count = y_count;
    rx.ycc = p.ycc ? 1 : 0;
    for (int c = 0; c < nc; c++) {
      const int pc = vc >= 0 ? vc : c;
      const bool up = vc < 0 && p.upsampling_path && p.upsampler[pc] && p.requested[pc];
      rx.wstart[c] = up ? p.wstart[pc] : 0;
      rx.wlimit[c] = up ? p.wlimit[pc] : (g.dnl && g.suby[c] > 1) ? g.blocks_h[c] * 8 : (g.height + g.suby[c] - 1) / g.suby[c];
    }
    rc = launch_reconstruct_ex(&b, d->stream, &rx);
    if (rc) return set_error(d, rc, rc == MIJPEG_ERR_DEVICE ? std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError())
                                                             : std::string("reconstruction not available for this request"));
  }
  return hand_out_request(d, p.min_x, p.min_y, y_count, cx1, cy1, min_comp, max_comp, nc, sb, row, padded, vc, all_on_view, to_device, dst, bpp, bpr);
} catch (...) { return boundary_catch(d, "mijpeg_display_rect"); }

// The scans of one codestream in the order the reference meets them: the codestream's own, then the ones that live in its
// refinement boxes.  A frame whose decode went through the sequential walk (damaged streams, DNL frames, the residual scan types of
// part 8) planned no scans: the walk's own record stands in (HostDecoder::walked_scans).
struct ScanStop { uint64_t begin, end; int32_t mcus_x, mcus_y; bool boxed; };
static void scans_of(const HostDecoder &h, bool all_boxed, std::vector<ScanStop> &out)
{
  if (h.walked()) {
    for (const auto &w : h.walked_scans()) out.push_back(ScanStop{(uint64_t)w.begin, (uint64_t)w.end, w.mcus_x, w.mcus_y, all_boxed || w.boxed});
    return;
  }
  for (const Scan &sc : h.scans)
    if (!sc.base) out.push_back(ScanStop{(uint64_t)sc.ecs_begin, (uint64_t)sc.ecs_end, sc.mcus_x, sc.mcus_y, all_boxed});
  for (const Scan &sc : h.scans)
    if (sc.base) out.push_back(ScanStop{0, 0, sc.mcus_x, sc.mcus_y, true});
}
static void all_scans(mijpeg_decoder *d, std::vector<ScanStop> &out)
{
  // scans of the codestream itself first; then -- JPEG XT -- the scans that live in boxes (hidden refinement scans of the legacy
  // frame, the residual codestream and its refinement scans): the reference parses them from memory streams while its input
  // stands at the marker behind the legacy frame's last scan (the EOI); behind them the alpha channel's: its own codestream (ALFA
  // box), that one's refinement boxes, its residual codestream (Image::ParseAlphaChannel / ParseResidualStream of the alpha image,
  // codestream/image.cpp:1337-1404, 1440-1462)
  scans_of(d->host, false, out);
  if (HostDecoder *res = d->host.residual()) scans_of(*res, true, out);
  if (d->alpha && d->alpha_ready) {
    scans_of(d->alpha->host, true, out);
    if (HostDecoder *ares = d->alpha->host.residual()) scans_of(*ares, true, out);
  }
}

int mijpeg_scan_offsets(mijpeg_decoder *d, uint64_t *first_byte, uint64_t *end_byte, int capacity)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  std::vector<ScanStop> all;
  all_scans(d, all);
  // (boxed scans: the input stands behind the last scan of the codestream itself; end 0 marks them)
  int n = 0;
  uint64_t behind = 0;
  for (const ScanStop &sc : all) {
    if (sc.boxed) continue;
    if (n < capacity) {
      if (first_byte) first_byte[n] = sc.begin;
      if (end_byte) end_byte[n] = sc.end;
    }
    behind = sc.end;
    n++;
  }
  for (const ScanStop &sc : all) {
    if (!sc.boxed) continue;
    if (n < capacity) {
      if (first_byte) first_byte[n] = behind;
      if (end_byte) end_byte[n] = 0;
    }
    n++;
  }
  return n;
} catch (...) { return boundary_catch(d, "mijpeg_scan_offsets"); }

int mijpeg_scan_grids(mijpeg_decoder *d, int32_t *mcus_x, int32_t *mcus_y, int capacity)
try {
  if (!d) return MIJPEG_ERR_INVALID_PARAMETER;
  std::vector<ScanStop> all;
  all_scans(d, all);
  int n = 0;
  for (int boxed = 0; boxed < 2; boxed++) // (the same order as mijpeg_scan_offsets)
    for (const ScanStop &sc : all) {
      if ((int)sc.boxed != boxed) continue;
      if (n < capacity) {
        if (mcus_x) mcus_x[n] = sc.mcus_x;
        if (mcus_y) mcus_y[n] = sc.mcus_y;
      }
      n++;
    }
  return n;
} catch (...) { return boundary_catch(d, "mijpeg_scan_grids"); }

int mijpeg_display_plan(mijpeg_decoder *d, int32_t min_x, int32_t min_y, int32_t max_x, int32_t max_y, int32_t min_comp, int32_t max_comp,
                        uint32_t flags, const uint32_t bm_height[MIJPEG_MAX_COMPONENTS], int32_t out[8 + 6 * MIJPEG_MAX_COMPONENTS])
try {
  if (!d || !bm_height || !out) return MIJPEG_ERR_INVALID_PARAMETER;
  if (d->host.info.components < 1 || d->host.info.width < 1) return set_error(d, MIJPEG_ERR_OBJECT_DOESNT_EXIST, "no parsed stream: call mijpeg_read_header first");
  const mijpeg_info &f = d->host.info;
  ensure_request_models(d);
  const RequestPlan p = d->model.request(min_x, min_y, max_x, max_y, min_comp, max_comp, !(flags & MIJPEG_FLAG_NO_UPSAMPLING),
                                         !(flags & MIJPEG_FLAG_NO_COLOR_TRANSFORM), bm_height);
  bool rplain = true;
  if (d->host.is_xt() && f.components == 3 && !d->host.xt.no_residual) // (the residual image: cursors through mijpeg_display_cursor(4 + c))
    rplain = d->rmodel.request(min_x, min_y, max_x, max_y, min_comp, max_comp, !(flags & MIJPEG_FLAG_NO_UPSAMPLING),
                               !(flags & MIJPEG_FLAG_NO_COLOR_TRANSFORM), bm_height).plain;
  out[0] = p.nothing; out[1] = p.plain && rplain; out[2] = p.ycc; out[3] = p.view;
  out[4] = p.min_x; out[5] = p.min_y; out[6] = p.max_x; out[7] = p.max_y;
  for (int c = 0; c < MIJPEG_MAX_COMPONENTS; c++) {
    int32_t *o = out + 8 + 6 * c;
    o[0] = d->model.cursor(c); o[1] = p.g0[c]; o[2] = p.g1[c]; o[3] = p.wstart[c]; o[4] = p.wlimit[c];
    int zeros = 0;
    for (int g = p.g0[c]; g <= p.g1[c] && g < (int)p.rowmap[c].size(); g++) zeros += p.rowmap[c][(size_t)g] < 0;
    o[5] = zeros;
  }
  return MIJPEG_OK;
} catch (...) { return boundary_catch(d, "mijpeg_display_plan"); }

int mijpeg_display_cursor(mijpeg_decoder *d, int component)
try {
  if (!d || component < 0 || component >= 2 * MIJPEG_MAX_COMPONENTS || !d->model_valid) return 0;
  return component >= MIJPEG_MAX_COMPONENTS ? d->rmodel.cursor(component - MIJPEG_MAX_COMPONENTS) : d->model.cursor(component);
} catch (...) { return boundary_catch(d, "mijpeg_display_cursor"); }

} // extern "C"
