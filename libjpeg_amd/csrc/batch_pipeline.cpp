// batch_pipeline.cpp -- BASELINE config 4's driver behind the C ABI: a batch of independent streams of one shape, bytes in host
// memory -> pixels in HBM, as a software pipeline over several decoder objects.
//
// The reference has no counterpart (it is single-image, single-threaded: cmd/reconstruct.cpp decodes one file); what is
// pipelined here are the library's own batch entry points, all of them public (include/mijpeg.h): one thread drives
// `decoder_objects` decoder objects round-robin,
//     submit(chunk i)       host: header parse, restart marker search, tables, gather into pinned memory; ENQUEUES upload + Huffman kernel
//     reconstruct(chunk i)  when the object comes round again: waits for what its Huffman kernel reported, launches the fused kernel
// so that the host prepares chunks i + 1, i + 2 while the copy engine and the compute units work on chunk i.  A chunk the device
// path declines (MIJPEG_ERR_NOT_AVAILABLE: a stream that does not qualify, a damaged one, a device walk that had not settled)
// is decoded by the blocking batch call, which watches the walk, and failing that image by image with the host entropy decoder;
// the reconstruction is the device's in every case.  Optionally every chunk's pixels start their way to (pinned) host memory as
// soon as the chunk's reconstruction is through, on a stream of the pipeline's own (mijpeg_stream_wait: no host wait).
//
// Rounds 2-5 had this loop in Python (libjpeg_amd/batch.py); a C or C++ client had to write it again.  batch.py is now a
// binding of these three calls.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <new>
#include <string>
#include <vector>

#include "../../include/mijpeg.h"

struct mijpeg_batch_pipeline {
  int device = 0, chunk = 32, ramp = 0;
  std::vector<mijpeg_decoder *> dec;
  hipStream_t download = nullptr;
  // diagnostics of the last run
  int32_t chunks = 0, fallbacks = 0, redone = 0;
  bool speculate = false, finished_once = false;
  std::vector<float> submit_ms;
  int err_code = 0;
  std::string err_msg;
};

namespace {
int fail(mijpeg_batch_pipeline *p, int code, const char *msg)
{
  p->err_code = code;
  try {
    p->err_msg = msg ? msg : "";
  } catch (...) {
    p->err_msg.clear();
  }
  return code;
}
int fail_from(mijpeg_batch_pipeline *p, mijpeg_decoder *d, int code)
{
  const char *m = nullptr;
  mijpeg_last_error(d, &m);
  return fail(p, code, m ? m : "decoder error");
}

// [first, last + 1) of the chunks.  ramp: chunk / 4, chunk / 2 in front, chunk / 2, chunk / 4 at the end -- the upload of the
// first chunk starts as soon as a few frames are parsed, and the kernels that trail the last upload are short (the link is the
// bound: what is not under an upload is the pipeline's fill and drain time).
void schedule(int n, int chunk, int ramp, std::vector<std::pair<int, int>> &out)
{
  std::vector<int> sizes;
  if (ramp && n >= 4 * chunk && chunk >= 8) {
    const int head[2] = {chunk / 4 > 2 ? chunk / 4 : 2, chunk / 2 > 4 ? chunk / 2 : 4};
    const int body = n - 2 * (head[0] + head[1]);
    const int nb = (body + chunk - 1) / chunk; // the body in equal chunks of at most `chunk` frames
    sizes.push_back(head[0]);
    sizes.push_back(head[1]);
    for (int i = 0; i < nb; i++) sizes.push_back(body / nb + (i < body % nb ? 1 : 0));
    sizes.push_back(head[1]);
    sizes.push_back(head[0]);
  } else {
    for (int i = 0; i < n / chunk; i++) sizes.push_back(chunk);
    if (n % chunk) sizes.push_back(n % chunk);
  }
  out.clear();
  int a = 0;
  for (int sz : sizes) {
    out.push_back(std::make_pair(a, a + sz));
    a += sz;
  }
}
} // namespace

extern "C" {

int mijpeg_batch_pipeline_create(mijpeg_batch_pipeline **out, int device, int chunk_frames, int decoder_objects, int ramp)
try {
  if (!out) return MIJPEG_ERR_INVALID_PARAMETER;
  *out = nullptr;
  if (device < 0 || chunk_frames < 1 || decoder_objects < 1 || decoder_objects > 16) return MIJPEG_ERR_INVALID_PARAMETER;
  mijpeg_batch_pipeline *p = new (std::nothrow) mijpeg_batch_pipeline();
  if (!p) return MIJPEG_ERR_OUT_OF_MEMORY;
  p->device = device;
  p->chunk = chunk_frames;
  p->ramp = ramp ? 1 : 0;
  for (int k = 0; k < decoder_objects; k++) {
    mijpeg_decoder *d = nullptr;
    const int rc = mijpeg_create(&d, device);
    if (rc) {
      for (mijpeg_decoder *e : p->dec) mijpeg_destroy(e);
      delete p;
      return rc;
    }
    p->dec.push_back(d);
  }
  *out = p;
  return MIJPEG_OK;
} catch (const std::bad_alloc &) {
  return MIJPEG_ERR_OUT_OF_MEMORY;
} catch (...) {
  return MIJPEG_ERR_PHASE_ERROR;
}

void mijpeg_batch_pipeline_destroy(mijpeg_batch_pipeline *p)
{
  if (!p) return;
  for (mijpeg_decoder *d : p->dec) mijpeg_destroy(d);
  if (p->download) {
    (void)hipSetDevice(p->device);
    (void)hipStreamSynchronize(p->download);
    (void)hipStreamDestroy(p->download);
  }
  delete p;
}

int mijpeg_batch_pipeline_schedule(int n, int chunk_frames, int ramp, int32_t *first, int32_t *end, int capacity)
try {
  if (n < 0 || chunk_frames < 1) return MIJPEG_ERR_INVALID_PARAMETER;
  std::vector<std::pair<int, int>> ch;
  schedule(n, chunk_frames, ramp, ch);
  for (size_t i = 0; i < ch.size() && (int)i < capacity; i++) {
    if (first) first[i] = ch[i].first;
    if (end) end[i] = ch[i].second;
  }
  return (int)ch.size();
} catch (...) {
  return MIJPEG_ERR_OUT_OF_MEMORY;
}

int mijpeg_batch_pipeline_run(mijpeg_batch_pipeline *p, const uint8_t *const *streams, const size_t *sizes, int n, void *dst_device,
                              int64_t frame_stride, int64_t row_stride, void *download_host)
try {
  if (!p || !streams || !sizes || n < 0 || !dst_device) return MIJPEG_ERR_INVALID_PARAMETER;
  p->err_code = 0;
  p->err_msg.clear();
  p->fallbacks = 0;
  p->redone = 0;
  p->chunks = 0;
  p->submit_ms.clear();
  if (n == 0) return MIJPEG_OK;
  if (hipSetDevice(p->device) != hipSuccess) return fail(p, MIJPEG_ERR_DEVICE, "hipSetDevice failed");
  std::vector<std::pair<int, int>> chunks;
  schedule(n, p->chunk, p->ramp, chunks);
  p->chunks = (int32_t)chunks.size();
  p->submit_ms.assign(chunks.size(), 0.f);
  const int depth = (int)p->dec.size();
  uint8_t *base = (uint8_t *)dst_device;
  if (download_host && !p->download && hipStreamCreateWithFlags(&p->download, hipStreamNonBlocking) != hipSuccess)
    return fail(p, MIJPEG_ERR_DEVICE, "no download stream");

  // Whatever happens below, nothing of this batch stays in flight behind the caller's back (the objects' kernels write into dst_device)
  struct Drain {
    mijpeg_batch_pipeline *p;
    ~Drain()
    {
      for (mijpeg_decoder *d : p->dec) (void)mijpeg_synchronize(d);
      if (p->download) (void)hipStreamSynchronize(p->download);
    }
  } drain{p};

  auto download = [&](int k, int a, int b) -> int {
    if (!download_host) return MIJPEG_OK;
    int rc = mijpeg_stream_wait(p->dec[(size_t)k], (void *)p->download);
    if (rc) return fail_from(p, p->dec[(size_t)k], rc);
    if (hipMemcpyAsync((uint8_t *)download_host + (int64_t)a * frame_stride, base + (int64_t)a * frame_stride, (size_t)((int64_t)(b - a) * frame_stride),
                       hipMemcpyDeviceToHost, p->download) != hipSuccess)
      return fail(p, MIJPEG_ERR_DEVICE, "download of a chunk's pixels failed");
    return MIJPEG_OK;
  };
  // what include/mijpeg.h prescribes when the device path declines a chunk
  auto one_by_one = [&](mijpeg_decoder *d, int a, int b) -> int {
    p->fallbacks++;
    int rc = mijpeg_decode_batch_device(d, streams + a, sizes + a, b - a, 1);
    if (rc == MIJPEG_OK) rc = mijpeg_reconstruct_batch_device(d, base + (int64_t)a * frame_stride, frame_stride, row_stride, 0, 1);
    if (rc == MIJPEG_OK) return MIJPEG_OK;
    if (rc != MIJPEG_ERR_NOT_AVAILABLE) return fail_from(p, d, rc);
    for (int i = a; i < b; i++) {
      rc = mijpeg_set_input(d, streams[i], sizes[i]);
      if (rc == MIJPEG_OK) rc = mijpeg_decode_coefficients(d, 0);
      if (rc == MIJPEG_OK) rc = mijpeg_reconstruct_device(d, base + (int64_t)i * frame_stride, row_stride, 0, 1);
      if (rc) return fail_from(p, d, rc);
    }
    return MIJPEG_OK;
  };
  std::vector<int> busy((size_t)depth, -1); // chunk whose Huffman kernel is on the object's stream
  // MIJPEG_FLAG_SPECULATIVE (opt-in, mijpeg_batch_pipeline_speculation): once a chunk of this material has been through -- its
  // range check is the library's hint -- the reconstruction of a chunk is launched right behind its Huffman kernel and validated
  // when the object comes round again; the thread never waits for the device inside the pipeline.  Measured in round 5: a wash
  // on the medians (the two kernels then run side by side and each takes twice its time), profiles/r05/batch4k_stall.txt.
  std::vector<int> unvalidated((size_t)depth, -1);
  auto validate = [&](int k) -> int {
    const int ci = unvalidated[(size_t)k];
    if (ci < 0) return MIJPEG_OK;
    unvalidated[(size_t)k] = -1;
    const int a = chunks[(size_t)ci].first, b = chunks[(size_t)ci].second;
    int rc = mijpeg_finish_batch_device(p->dec[(size_t)k]);
    if (rc == MIJPEG_OK) {
      if (mijpeg_batch_speculation(p->dec[(size_t)k], nullptr, nullptr) == 1) { // the assumed range check did not hold: the pixels were made again
        p->redone++;
        return download(k, a, b);
      }
      return MIJPEG_OK;
    }
    if (rc != MIJPEG_ERR_NOT_AVAILABLE) return fail_from(p, p->dec[(size_t)k], rc);
    rc = one_by_one(p->dec[(size_t)k], a, b);
    return rc ? rc : download(k, a, b);
  };
  auto retire = [&](int k) -> int {
    const int a = chunks[(size_t)busy[(size_t)k]].first, b = chunks[(size_t)busy[(size_t)k]].second;
    busy[(size_t)k] = -1;
    int rc = mijpeg_reconstruct_batch_device(p->dec[(size_t)k], base + (int64_t)a * frame_stride, frame_stride, row_stride, 0, 0);
    if (rc == MIJPEG_ERR_NOT_AVAILABLE) rc = one_by_one(p->dec[(size_t)k], a, b);
    else if (rc) return fail_from(p, p->dec[(size_t)k], rc);
    if (rc) return rc;
    p->finished_once = true;
    return download(k, a, b);
  };
  for (size_t ci = 0; ci < chunks.size(); ci++) {
    const int k = (int)(ci % (size_t)depth);
    mijpeg_decoder *d = p->dec[(size_t)k];
    if (busy[(size_t)k] >= 0) {
      const int rc = retire(k);
      if (rc) return rc;
    }
    {
      const int vrc = validate(k);
      if (vrc) return vrc;
    }
    const int a = chunks[ci].first, b = chunks[ci].second;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = mijpeg_submit_batch_device(d, streams + a, sizes + a, b - a, 1);
    if (rc == MIJPEG_OK && p->speculate && p->finished_once) {
      rc = mijpeg_reconstruct_batch_device(d, base + (int64_t)a * frame_stride, frame_stride, row_stride, MIJPEG_FLAG_SPECULATIVE, 0);
      if (rc == MIJPEG_OK) {
        unvalidated[(size_t)k] = (int)ci;
        rc = download(k, a, b);
        if (rc) return rc;
      }
    } else if (rc == MIJPEG_OK)
      busy[(size_t)k] = (int)ci;
    if (rc == MIJPEG_ERR_NOT_AVAILABLE) {
      rc = one_by_one(d, a, b);
      if (rc == MIJPEG_OK) rc = download(k, a, b);
      if (rc) return rc;
    } else if (rc != MIJPEG_OK)
      return fail_from(p, d, rc);
    p->submit_ms[ci] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  // what is still on the device, oldest chunk first
  for (;;) {
    int k = -1;
    for (int j = 0; j < depth; j++)
      if (busy[(size_t)j] >= 0 && (k < 0 || busy[(size_t)j] < busy[(size_t)k])) k = j;
    if (k < 0) break;
    const int rc = retire(k);
    if (rc) return rc;
  }
  for (int k = 0; k < depth; k++) {
    const int rc = validate(k);
    if (rc) return rc;
  }
  for (mijpeg_decoder *d : p->dec) {
    const int rc = mijpeg_synchronize(d);
    if (rc) return fail_from(p, d, rc);
  }
  if (p->download && hipStreamSynchronize(p->download) != hipSuccess) return fail(p, MIJPEG_ERR_DEVICE, "download stream");
  return MIJPEG_OK;
} catch (const std::bad_alloc &) {
  return p ? fail(p, MIJPEG_ERR_OUT_OF_MEMORY, "mijpeg_batch_pipeline_run: out of memory") : MIJPEG_ERR_OUT_OF_MEMORY;
} catch (...) {
  return p ? fail(p, MIJPEG_ERR_PHASE_ERROR, "mijpeg_batch_pipeline_run: unexpected exception") : MIJPEG_ERR_PHASE_ERROR;
}

int mijpeg_batch_pipeline_stats(mijpeg_batch_pipeline *p, int32_t *chunks, int32_t *fallbacks, float *submit_ms, int capacity)
{
  if (!p) return MIJPEG_ERR_INVALID_PARAMETER;
  if (chunks) *chunks = p->chunks;
  if (fallbacks) *fallbacks = p->fallbacks;
  for (size_t i = 0; submit_ms && i < p->submit_ms.size() && (int)i < capacity; i++) submit_ms[i] = p->submit_ms[i];
  return MIJPEG_OK;
}

int mijpeg_batch_pipeline_speculation(mijpeg_batch_pipeline *p, int on)
{
  if (!p) return MIJPEG_ERR_INVALID_PARAMETER;
  if (on >= 0) p->speculate = on != 0;
  return p->redone; // chunks of the last run whose speculative reconstruction had to be made again
}

int mijpeg_batch_pipeline_last_error(mijpeg_batch_pipeline *p, const char **message)
{
  if (!p) return MIJPEG_ERR_INVALID_PARAMETER;
  if (message) *message = p->err_code ? p->err_msg.c_str() : nullptr;
  return p->err_code;
}

mijpeg_decoder *mijpeg_batch_pipeline_decoder(mijpeg_batch_pipeline *p, int k)
{
  return p && k >= 0 && k < (int)p->dec.size() ? p->dec[(size_t)k] : nullptr;
}

} // extern "C"
