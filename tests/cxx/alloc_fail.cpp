// Exception firewall of the C boundary ("nothing propagates as a C++ exception"; the reference turns every failure,
// memory included, into an error code: tools/environment.hpp:752-784, interface/jpeg.cpp:205-220).
//
// A client of include/mijpeg.h ONLY (plus the source-compatible class JPEG for the second half) that replaces the global operator
// new with one that fails -- throws std::bad_alloc -- at the N-th allocation after it is armed, for N = 1, 2, 3, ... until a whole
// decode goes through: every call must come back with MIJPEG_OK or an error code (out of memory among them), never with an
// exception on the client's side of the boundary and never with a terminated process.  Host only (device -1): parse + entropy
// decode are where the library's own containers grow.
//
//   alloc_fail <stream.jpg> [threads] [stride]      prints one summary line; exit code 0 = the firewall held
#include <atomic>
#include <exception>
#include <execinfo.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "../../include/mijpeg.h"
#include "interface/hooks.hpp"
#include "interface/jpeg.hpp"
#include "interface/parameters.hpp"
#include "interface/tagitem.hpp"

static std::atomic<long> g_countdown{-1}; // < 0: not armed
static std::atomic<long> g_failures{0};

static void *counted_alloc(std::size_t n)
{
  if (g_countdown.load(std::memory_order_relaxed) >= 0 && g_countdown.fetch_sub(1) == 0) {
    g_failures.fetch_add(1);
    throw std::bad_alloc();
  }
  void *p = malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void *operator new(std::size_t n) { return counted_alloc(n); }
void *operator new[](std::size_t n) { return counted_alloc(n); }
void *operator new(std::size_t n, const std::nothrow_t &) noexcept
{
  try {
    return counted_alloc(n);
  } catch (...) {
    return nullptr;
  }
}
void *operator new[](std::size_t n, const std::nothrow_t &) noexcept
{
  try {
    return counted_alloc(n);
  } catch (...) {
    return nullptr;
  }
}
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, std::size_t) noexcept { free(p); }
void operator delete[](void *p, std::size_t) noexcept { free(p); }

struct Source {
  const uint8_t *data;
  size_t size, pos;
};
static JPG_LONG io_hook(struct JPG_Hook *hook, struct JPG_TagItem *tags)
{
  Source *s = (Source *)hook->hk_pData;
  if (tags->GetTagData(JPGTAG_FIO_ACTION) != JPGFLAG_ACTION_READ) return -1;
  uint8_t *buf = (uint8_t *)tags->GetTagPtr(JPGTAG_FIO_BUFFER);
  size_t n = (size_t)tags->GetTagData(JPGTAG_FIO_SIZE);
  if (n > s->size - s->pos) n = s->size - s->pos;
  for (size_t i = 0; i < n; i++) buf[i] = s->data[s->pos + i];
  s->pos += n;
  return (JPG_LONG)n;
}

static void on_terminate()
{ // where an exception got away: the frames, for addr2line
  void *frames[48];
  g_countdown.store(-1);
  backtrace_symbols_fd(frames, backtrace(frames, 48), 2);
  abort();
}

int main(int argc, char **argv)
{
  std::set_terminate(on_terminate);
  if (argc < 2) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  static uint8_t buf[1 << 24];
  const size_t size = fread(buf, 1, sizeof(buf), f);
  fclose(f);
  const int threads = argc > 2 ? atoi(argv[2]) : 1;
  const long stride = argc > 3 ? atol(argv[3]) : 1;

  // ---- the C ABI: create / set_input / decode_coefficients / get_info / destroy
  long ok_at = -1, oom = 0, other = 0;
  for (long n = 0; n < 2000000 && ok_at < 0; n += stride) {
    mijpeg_decoder *d = nullptr;
    g_countdown.store(n);
    int rc = mijpeg_create(&d, -1);
    if (rc == MIJPEG_OK) rc = mijpeg_set_input(d, buf, size);
    if (rc == MIJPEG_OK) rc = mijpeg_decode_coefficients(d, threads);
    mijpeg_info info;
    if (rc == MIJPEG_OK) rc = mijpeg_get_info(d, &info);
    const bool tripped = g_countdown.load() < 0;
    g_countdown.store(-1);
    if (rc != MIJPEG_OK && d) { // the error is readable, and is the object's own
      const char *msg = nullptr;
      if (mijpeg_last_error(d, &msg) != rc && rc != MIJPEG_ERR_OUT_OF_MEMORY) { printf("FAIL last_error disagrees at %ld: %d\n", n, rc); return 1; }
    }
    if (d) mijpeg_destroy(d);
    if (rc == MIJPEG_OK && !tripped) ok_at = n; // (the allocation that would have failed was never reached)
    else if (rc == MIJPEG_ERR_OUT_OF_MEMORY) oom++;
    else if (rc != MIJPEG_OK) other++;
  }
  if (ok_at < 0) { printf("FAIL the decode never got through\n"); return 1; }

  // ---- class JPEG on top of it: Construct / Read
  long jok_at = -1, joom = 0, jother = 0;
  for (long n = 0; n < 2000000 && jok_at < 0; n += stride) {
    g_countdown.store(n);
    Source src{buf, size, 0};
    struct JPG_Hook hook(io_hook, &src);
    JPEG *j = nullptr;
    {
      struct JPG_TagItem ctags[] = {JPG_ValueTag(JPGTAG_MIJPEG_DEVICE, -1), JPG_EndTag};
      j = JPEG::Construct(ctags);
    }
    JPG_LONG okay = JPG_FALSE;
    int code = JPGERR_OUT_OF_MEMORY; // (a Construct that returns NULL is the reference's answer to "no memory" as well)
    if (j) {
      struct JPG_TagItem tags[] = {JPG_PointerTag(JPGTAG_HOOK_IOHOOK, &hook), JPG_PointerTag(JPGTAG_HOOK_IOSTREAM, nullptr), JPG_EndTag};
      okay = j->Read(tags);
      if (!okay) {
        const char *msg = nullptr;
        code = (int)j->LastError(msg);
      }
    }
    const bool tripped = g_countdown.load() < 0;
    g_countdown.store(-1);
    if (j) JPEG::Destruct(j);
    if (okay && !tripped) jok_at = n;
    else if (!okay && code == JPGERR_OUT_OF_MEMORY) joom++;
    else if (!okay) jother++;
  }
  if (jok_at < 0) { printf("FAIL JPEG::Read never got through\n"); return 1; }
  printf("OK c_abi: through at %ld allocations, %ld x out of memory, %ld x other code; class JPEG: through at %ld, %ld x out of memory, %ld x other; %ld injected failures\n",
         ok_at, oom, other, jok_at, joom, jother, g_failures.load());
  return 0;
}
