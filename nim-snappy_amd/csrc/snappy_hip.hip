// snappy_hip.hip -- host side of the C ABI declared in include/snappy_hip.h.
//
// Mirrors the reference's in-memory API (snappy.nim) over the gfx950 kernels.  What runs on
// the host here is only what the reference itself treats as scalar bookkeeping: size bounds,
// varint length readers and the framed-stream header walk (codec.nim:92-214, the chunk loop
// of snappy.nim:199-265).  Block encode, block decode and CRC32C always run on the GPU; if
// HIP is unusable every codec entry point returns SNAPPY_HIP_DEVICE_ERROR.
#include "../../include/snappy_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <string>
#include <functional>
#include <vector>

#include "common.h"
#include "crc_pack_kernels.h"
#include "decode2_kernel.h"
#include "decode_kernel.h"
#include "encode_kernel.h"
#include "framed_kernels.h"
#include "index_kernel.h"
#include "split_kernels.h"

using namespace snappy_hip;

namespace {

thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
      return SNAPPY_HIP_DEVICE_ERROR;                                                        \
    }                                                                                        \
  } while (0)

// Debug knobs (profiles/README.md) are read only by builds with -DSNAPPY_HIP_DEBUG.  The shipped library reads
// four tuning knobs of the host-buffer calls, once (include/snappy_hip.h documents them): SNAPPY_HIP_DEVICE,
// SNAPPY_HIP_HOST_BATCH, SNAPPY_HIP_PIN_HOST, SNAPPY_HIP_COPY_THREADS; every context reads SNAPPY_HIP_ENC_GWAVES when it is created.
#ifdef SNAPPY_HIP_DEBUG
inline const char* dbg_env(const char* name) { return getenv(name); }
#else
inline const char* dbg_env(const char*) { return nullptr; }
#endif

// every launch is checked where it is made
#define LAUNCH(...)                   \
  do {                                \
    hipLaunchKernelGGL(__VA_ARGS__);  \
    HIP_TRY(hipGetLastError());       \
  } while (0)

constexpr uint32_t kSlotStride = 76800;  // >= 8 + 3 + 76490, multiple of 256
const uint8_t kFramingHeader[10] = {0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59};

// ---- LEB128 (stew/leb128; call sites snappy.nim:49,92, codec.nim:134) ----------------------
int varint_decode(const uint8_t* in, size_t n, int bits, uint64_t* val) {
  const int max_len = (bits + 6) / 7;
  uint64_t v = 0;
  for (int i = 0; i < max_len && (size_t)i < n; i++) {
    uint8_t b = in[i];
    if (i == max_len - 1 && (b >> (bits - 7 * i))) return 0;
    v |= (uint64_t)(b & 0x7f) << (7 * i);
    if (!(b & 0x80)) {
      *val = v;
      return i + 1;
    }
  }
  return 0;
}

int varint_encode_u32(uint32_t v, uint8_t* out) {
  int i = 0;
  while (v >= 0x80) {
    out[i++] = (uint8_t)(v | 0x80);
    v >>= 7;
  }
  out[i++] = (uint8_t)v;
  return i;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// Makes a context's device current for one call and gives the caller's device back afterwards (a
// thread's current device is the caller's state: PyTorch's, or another context's).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

}  // namespace

constexpr int kTimeSlots = 10;  // snappy_hip_ctx_kernel_ms(which)
struct snappy_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side_stream = nullptr;  // work that runs beside the main stream's (a framed stream's stored chunks)
  hipEvent_t side_done = nullptr;
  // One stream per context at a time (include/snappy_hip.h): the workspace, the encoder's work queue and the counters are
  // the context's.  A call on another stream than the call before it first waits for that call's work (StreamTurn, below).
  hipStream_t last_stream = nullptr;
  hipEvent_t last_done = nullptr;
  bool last_pending = false;
  uint32_t* d_crc_tab = nullptr;   // [4][256]
  uint32_t* d_col_mul = nullptr;   // [256]
  uint16_t* d_tag_lut = nullptr;   // [256] the decode front end's tag table (decode2_kernel.h)
  uint32_t crc_k32k = 0;           // x^(8 * 32768) mod P (decode2_kernel.h)
  uint32_t* d_seq_off = nullptr;   // [kSeqLen]
  uint32_t* d_seq_step = nullptr;  // [kSeqLen]
  uint32_t* d_counters = nullptr;  // [16] [0] turns the indexed decoder gave up on (kernel_ms slot 9); [2..3] a 64-bit sum:
                                   // stream + output bytes of the units the index pass decoded itself, [4] how many (slot 10)
  DevBuf ws[24];                   // grow-only workspace of the host-buffer API
  // page-locked staging ring of the host-buffer calls (stage_*, below): kStageSlots pieces, an event each
  uint8_t* stage = nullptr;
  hipEvent_t stage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool stage_busy[4] = {false, false, false, false};
  bool stage_failed = false;       // the ring could not be allocated: copies go the runtime's pageable way
  int n_cus = 256;                 // compute units of the device (the encoder's persistent grid: four workgroups each)
  uint64_t enc_g_min_blocks = 0;   // ... from this many blocks a batch on (0: twice the resident workgroups)
  uint32_t enc_g_per4 = 4;         // of four encoder workgroups, how many run a second wave with its table in global memory
  bool launch_order = true;        // batches of >= 512 units are launched in sorted order (snappy_hip_ctx_launch_order)
  bool timing = false;
  struct Timed {
    hipEvent_t a, b;
    int which;
  };
  std::vector<Timed> timed;
  double ms_sum[kTimeSlots] = {};
  uint64_t ms_cnt[kTimeSlots] = {};
};

namespace {

int ws_get(snappy_hip_ctx* c, int slot, size_t bytes, void** out) {
  DevBuf& b = c->ws[slot];
  if (b.cap < bytes) {
    if (b.p) HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 4096;
    HIP_TRY(hipMalloc(&b.p, want));
    b.cap = want;
  }
  *out = b.p;
  return SNAPPY_HIP_OK;
}

extern "C" double snappy_hip_ctx_kernel_ms(snappy_hip_ctx* c, int which, uint64_t* launches);

struct LaunchTimer {
  snappy_hip_ctx* c;
  hipStream_t s;
  int which;
  hipEvent_t a = nullptr, b = nullptr;
  LaunchTimer(snappy_hip_ctx* c_, hipStream_t s_, int w) : c(c_), s(s_), which(w) {
    if (c->timing && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
      (void)hipEventRecord(a, s);
  }
  ~LaunchTimer() {
    if (a && b) {
      (void)hipEventRecord(b, s);
      c->timed.push_back({a, b, which});
      if (c->timed.size() >= 4096) (void)snappy_hip_ctx_kernel_ms(c, -1, nullptr);  // fold them into the sums
    }
  }
};

hipStream_t pick_stream(snappy_hip_ctx* c, void* stream) {
  return stream ? (hipStream_t)stream : c->stream;
}

// A device-resident call's turn on its context: work that an earlier call left pending on ANOTHER stream is waited for on
// this one before anything is enqueued (the two calls would otherwise share the workspace slots and the encoder's queue
// counter -- skipped blocks, stale sizes), and the end of this call's work is marked for the next.  Same stream: nothing
// to do, a stream is in order.
struct StreamTurn {
  snappy_hip_ctx* c;
  hipStream_t s;
  StreamTurn(snappy_hip_ctx* c_, hipStream_t s_) : c(c_), s(s_) {
    if (c->last_pending && c->last_stream != s && c->last_done) (void)hipStreamWaitEvent(s, c->last_done, 0);
  }
  ~StreamTurn() {
    if (c->last_done && hipEventRecord(c->last_done, s) == hipSuccess) {
      c->last_stream = s;
      c->last_pending = true;
    }
  }
  StreamTurn(const StreamTurn&) = delete;
  StreamTurn& operator=(const StreamTurn&) = delete;
};

// perm (in workspace slot `slot`, after `head` bytes the caller keeps there) = the n items in the
// order of their keys' buckets (crc_pack_kernels.h): a counting sort in three small launches.
int launch_order(snappy_hip_ctx* c, const uint32_t* d_keys, uint64_t n, int mode, int slot, size_t head,
                 hipStream_t s, void** d_perm) {
  void* base;
  int st = ws_get(c, slot, head + n * 4 + kOrderBuckets * 4, &base);
  if (st) return st;
  uint32_t* perm = (uint32_t*)((uint8_t*)base + head);
  uint32_t* counts = perm + n;
  HIP_TRY(hipMemsetAsync(counts, 0, kOrderBuckets * 4, s));
  const uint32_t gb = (uint32_t)((n + 1023) / 1024 < 256 ? (n + 1023) / 1024 : 256);
  LAUNCH(order_count_kernel, dim3(gb), dim3(256), 0, s, d_keys, n, mode, counts);
  LAUNCH(order_scan_kernel, dim3(1), dim3(64), 0, s, counts);
  LAUNCH(order_scatter_kernel, dim3(gb), dim3(256), 0, s, d_keys, n, mode, counts, perm);
  *d_perm = perm;
  return SNAPPY_HIP_OK;
}

// CRC tables (crc_pack_kernels.h): generated, the reference's tables are not copied.
void build_crc_tables(uint32_t* stride_tab, uint32_t* col_mul) {
  uint32_t t0[256];
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? kCrcPoly : 0);
    t0[i] = c;
  }
  for (uint32_t k = 0; k < 4; k++)
    for (uint32_t b = 0; b < 256; b++) {
      uint32_t c = b << (8 * k);
      for (uint32_t z = 0; z < 4 * kCrcThreads; z++) c = t0[c & 0xff] ^ (c >> 8);
      stride_tab[k * 256 + b] = c;
    }
  for (uint32_t t = 0; t < kCrcThreads; t++) {
    uint32_t p = 0x80000000u;  // x^0
    for (uint32_t i = 0; i < 32 * (kCrcThreads - t); i++) p = (p >> 1) ^ ((p & 1) ? kCrcPoly : 0);
    col_mul[t] = p;
  }
}

// The reference's probe sequence (encoder.nim:263-331): skip starts at 32, step = skip >> 5.
void build_probe_sequence(uint32_t* off, uint32_t* step) {
  uint32_t skip = 32, o = 0;
  for (uint32_t j = 0; j < kSeqLen; j++) {
    step[j] = skip >> 5;
    off[j] = o;
    o += step[j];
    skip += step[j];
    if (o > (1u << 20)) o = 1u << 20;  // far beyond any block; keep it from wrapping
  }
}

}  // namespace

// =============================================================================================
// context
// =============================================================================================
extern "C" const char* snappy_hip_last_error(void) { return g_last_error.c_str(); }

namespace {
int ctx_init(snappy_hip_ctx* c) {
  {
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    if (cus > 0) c->n_cus = cus;
    if (const char* e = getenv("SNAPPY_HIP_ENC_GWAVES")) {  // tuning knob (include/snappy_hip.h): "g" or "g,min_blocks"
      const int v = atoi(e);
      c->enc_g_per4 = (uint32_t)(v < 0 ? 0 : (v > 4 ? 4 : v));
      if (const char* comma = strchr(e, ',')) c->enc_g_min_blocks = strtoull(comma + 1, nullptr, 10);
    }
  }
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  {  // (least priority: the main stream's small launches are not to queue behind the side stream's workgroups)
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, least));
  }
  HIP_TRY(hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming));
  // the indexed decoder's output window is dynamic LDS beyond the 64 KiB default limit
  HIP_TRY(hipFuncSetAttribute((const void*)decode_indexed_kernel<kMaxBlockLen>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)out_alloc(kMaxBlockLen) + 8192));
  std::vector<uint32_t> tab(1024), mul(kCrcThreads), so(kSeqLen), ss(kSeqLen);
  build_crc_tables(tab.data(), mul.data());
  {
    uint32_t p = 0x80000000u;  // x^0
    for (uint32_t i = 0; i < 8 * 32768; i++) p = (p >> 1) ^ ((p & 1) ? kCrcPoly : 0);
    c->crc_k32k = p;
  }
  build_probe_sequence(so.data(), ss.data());
  {
    // what a tag byte says about its element (decoder.nim:42-109): [0:7) length of the forms without length bytes,
    // [8:11) a copy1's offset bits 8..10, bit 11 literal, bit 12 literal with length bytes, bit 13 copy4
    uint16_t lut[256];
    for (uint32_t tg = 0; tg < 256; tg++) {
      const uint32_t hi6 = tg >> 2, ty = tg & 3;
      const uint32_t len = ty == 1 ? 4 + (hi6 & 7) : 1 + hi6;
      lut[tg] = (uint16_t)(len | (ty == 1 ? (tg & 0xe0u) << 3 : 0u) | (ty == 0 ? 0x800u : 0u) |
                           ((ty == 0 && hi6 >= 60) ? 0x1000u : 0u) | (ty == 3 ? 0x2000u : 0u));
    }
    HIP_TRY(hipMalloc((void**)&c->d_tag_lut, sizeof(lut)));
    HIP_TRY(hipMemcpy(c->d_tag_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc((void**)&c->d_crc_tab, tab.size() * 4));
  HIP_TRY(hipMalloc((void**)&c->d_col_mul, mul.size() * 4));
  HIP_TRY(hipMalloc((void**)&c->d_seq_off, so.size() * 4));
  HIP_TRY(hipMalloc((void**)&c->d_seq_step, ss.size() * 4));
  HIP_TRY(hipMemcpy(c->d_crc_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_col_mul, mul.data(), mul.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_seq_off, so.data(), so.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_seq_step, ss.data(), ss.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&c->d_counters, 64));
  HIP_TRY(hipMemset(c->d_counters, 0, 64));
  return SNAPPY_HIP_OK;
}
}  // namespace

extern "C" int snappy_hip_ctx_create(snappy_hip_ctx** out, int device) {
  *out = nullptr;
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) {
    g_last_error = "no such HIP device";
    return SNAPPY_HIP_DEVICE_ERROR;
  }
  DeviceGuard guard(device);  // (the caller's current device is the caller's again on return)
  snappy_hip_ctx* c = new snappy_hip_ctx();
  c->device = device;
  const int st = ctx_init(c);
  if (st) {
    snappy_hip_ctx_destroy(c);  // (tolerates a partly built context)
    return st;
  }
  *out = c;
  return SNAPPY_HIP_OK;
}

extern "C" void snappy_hip_ctx_destroy(snappy_hip_ctx* c) {
  if (!c) return;
  DeviceGuard guard(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& t : c->timed) {
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  for (auto& b : c->ws)
    if (b.p) (void)hipFree(b.p);
  if (c->stage) (void)hipHostFree(c->stage);
  for (auto& e : c->stage_ev)
    if (e) (void)hipEventDestroy(e);
  (void)hipFree(c->d_crc_tab);
  (void)hipFree(c->d_col_mul);
  (void)hipFree(c->d_tag_lut);
  (void)hipFree(c->d_seq_off);
  (void)hipFree(c->d_seq_step);
  (void)hipFree(c->d_counters);
  if (c->side_done) (void)hipEventDestroy(c->side_done);
  if (c->last_done) (void)hipEventDestroy(c->last_done);
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int snappy_hip_ctx_sync(snappy_hip_ctx* c, void* stream) {
  DeviceGuard guard(c->device);
  HIP_TRY(hipStreamSynchronize(pick_stream(c, stream)));
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_ctx_launch_order(snappy_hip_ctx* c, int enable) {
  c->launch_order = enable != 0;
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_ctx_timing(snappy_hip_ctx* c, int enable) {
  c->timing = enable != 0;
  if (enable) {
    for (auto& t : c->timed) {
      (void)hipEventDestroy(t.a);
      (void)hipEventDestroy(t.b);
    }
    c->timed.clear();
    for (int i = 0; i < kTimeSlots; i++) {
      c->ms_sum[i] = 0;
      c->ms_cnt[i] = 0;
    }
  }
  return SNAPPY_HIP_OK;
}

extern "C" double snappy_hip_ctx_kernel_ms(snappy_hip_ctx* c, int which, uint64_t* launches) {
  DeviceGuard guard(c->device);
  for (auto& t : c->timed) {
    float ms = 0;
    if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
      c->ms_sum[t.which] += ms;
      c->ms_cnt[t.which] += 1;
    }
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  c->timed.clear();
  if (which == 9) {  // not a duration: turns the indexed decoder gave up on since the context was created
    uint32_t v = 0;
    if (hipMemcpy(&v, c->d_counters, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
    if (launches) *launches = v;
    return (double)v;
  }
  if (which == 10) {  // not a duration either: bytes (stream + output) of the units the index pass decoded itself, and how many
    uint32_t v[3] = {0, 0, 0};
    if (hipMemcpy(v, c->d_counters + 2, 12, hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
    if (launches) *launches = v[2];
    return (double)(((uint64_t)v[1] << 32) | v[0]);
  }
  if (which < 0 || which >= kTimeSlots) return 0;
  if (launches) *launches = c->ms_cnt[which];
  return c->ms_cnt[which] ? c->ms_sum[which] / (double)c->ms_cnt[which] : 0.0;
}

// =============================================================================================
// device batch API
// =============================================================================================
extern "C" int snappy_hip_crc32c_d(snappy_hip_ctx* c, const uint8_t* d_in, const uint64_t* d_off,
                                   const uint32_t* d_len, uint64_t n_units, uint32_t* d_crc,
                                   void* stream) {
  if (n_units == 0) return SNAPPY_HIP_OK;
  DeviceGuard guard(c->device);
  CrcParams p{};
  p.in = d_in;
  p.off = d_off;
  p.len = d_len;
  p.crc = d_crc;
  p.n_units = n_units;
  p.stride_tab = c->d_crc_tab;
  p.col_mul = c->d_col_mul;
  hipStream_t s = pick_stream(c, stream);
  StreamTurn turn(c, s);
  {
    LaunchTimer lt(c, s, 2);
    LAUNCH(crc32c_units_kernel, dim3((uint32_t)n_units), dim3(kCrcThreads), 0, s, p);
  }
  HIP_TRY(hipGetLastError());
  return SNAPPY_HIP_OK;
}

namespace {
int crc_fixed_d(snappy_hip_ctx* c, const uint8_t* d_in, uint64_t total_len, uint32_t block_len,
                uint32_t* d_crc, hipStream_t s) {
  uint64_t nb = (total_len + block_len - 1) / block_len;
  if (nb == 0) return SNAPPY_HIP_OK;
  CrcParams p{};
  p.in = d_in;
  p.crc = d_crc;
  p.n_units = nb;
  p.stride_tab = c->d_crc_tab;
  p.col_mul = c->d_col_mul;
  p.total_len = total_len;
  p.block_len = block_len;
  {
    LaunchTimer lt(c, s, 2);
    LAUNCH(crc32c_units_kernel, dim3((uint32_t)nb), dim3(kCrcThreads), 0, s, p);
  }
  HIP_TRY(hipGetLastError());
  return SNAPPY_HIP_OK;
}
}  // namespace

namespace {
inline uint64_t host_batch_blocks() {  // SNAPPY_HIP_HOST_BATCH (read once): blocks per batch
  static const uint64_t v = [] {
    const char* e = getenv("SNAPPY_HIP_HOST_BATCH");
    const long x = e ? atol(e) : 0;
    return (uint64_t)(x >= 64 && x <= 65536 ? x : 2048);  // (measured: INTEGRATION.md)
  }();
  return v;
}
}  // namespace

extern "C" int snappy_hip_encode_blocks_d(snappy_hip_ctx* c, const uint8_t* d_in,
                                          uint64_t total_len, uint32_t block_len, int unit,
                                          uint8_t* d_slots, uint32_t slot_stride,
                                          uint32_t* d_sizes, void* stream) {
  if (block_len == 0 || block_len > kMaxBlockLen || unit < 0 || unit > 2 ||
      (uint64_t)slot_stride < snappy_hip_max_compressed_len(block_len) + 8) {
    return SNAPPY_HIP_INVALID_INPUT;
  }
  uint64_t nb = (total_len + block_len - 1) / block_len;
  if (nb == 0) return SNAPPY_HIP_OK;
  if (nb > 0x7fffffffull) return SNAPPY_HIP_INVALID_INPUT;
  DeviceGuard guard(c->device);
  hipStream_t s = pick_stream(c, stream);
  StreamTurn turn(c, s);
  uint32_t* d_crc = nullptr;
  if (unit == kUnitFrame) {
    void* p;
    int st = ws_get(c, 10, nb * 4, &p);
    if (st) return st;
    d_crc = (uint32_t*)p;
    st = crc_fixed_d(c, d_in, total_len, block_len, d_crc, s);
    if (st) return st;
  }
  EncodeParams p{};
  p.in = d_in;
  p.total_len = total_len;
  p.block_len = block_len;
  p.unit = unit;
  p.slots = d_slots;
  p.slot_stride = slot_stride;
  p.sizes = d_sizes;
  p.n_blocks = nb;
  p.crc = d_crc;
  p.seq_off = c->d_seq_off;
  p.seq_step = c->d_seq_step;
  if (nb >= 512 && c->launch_order && !dbg_env("SNAPPY_HIP_NO_ORDER")) {  // launch order: blocks that look alike together
    void *d_sk, *d_perm;
    int st = ws_get(c, 16, nb * 8 + kOrderBuckets * 4, &d_sk);
    if (st) return st;
    LAUNCH(encode_sketch_kernel, dim3((uint32_t)nb), dim3(64), 0, s, d_in, total_len, block_len, nb,
                       (uint32_t*)d_sk);
    if ((st = launch_order(c, (const uint32_t*)d_sk, nb, kOrderBySketch, 16, nb * 4, s, &d_perm))) return st;
    p.order = (const uint32_t*)d_perm;
  }
  unsigned long long* d_estats = nullptr;
  if (dbg_env("SNAPPY_HIP_STATS")) {  // DEBUG
    HIP_TRY(hipMalloc((void**)&d_estats, 128));
    HIP_TRY(hipMemsetAsync(d_estats, 0, 128, s));
    p.stats = d_estats;
  }
  {
    // Persistent workgroups (encode_kernel.h): four fit a CU (LDS), each takes blocks from one queue until none is left.
    // Their second waves -- tables in global memory, 32 KiB each, which must stay in the XCD's L2 -- run in g_per4 of every
    // four workgroups; a batch that does not fill the GPU's LDS-table waves starts none.
    const uint32_t resident = 4u * (uint32_t)c->n_cus;
    const uint32_t grid = nb < resident ? (uint32_t)nb : resident;
    // (from twice the resident workgroups on -- but never above a host-buffer call's batch, whatever the part's CU count:
    // the two defaults are one expression)
    const uint64_t g_min = c->enc_g_min_blocks ? c->enc_g_min_blocks
                                               : (2 * (uint64_t)resident < host_batch_blocks() ? 2 * (uint64_t)resident : host_batch_blocks());
    uint32_t g_per4 = nb >= g_min ? c->enc_g_per4 : 0;
    void* qp;
    int st = ws_get(c, 21, 64 + (g_per4 ? (size_t)grid * kMaxTableSize * 2 : 0), &qp);
    if (st) return st;
    HIP_TRY(hipMemsetAsync(qp, 0, 64, s));
    p.queue = (uint32_t*)qp;
    p.gtables = g_per4 ? (uint16_t*)((uint8_t*)qp + 64) : nullptr;
    p.g_per4 = g_per4;
    if (dbg_env("SNAPPY_HIP_ENC_DBG")) {  // DEBUG: bit 8 of g_per4 = the second waves alone; counters of who took how many
      p.dbg = 1;
      p.g_per4 |= (uint32_t)atoi(dbg_env("SNAPPY_HIP_ENC_DBG")) & 0x100u;
    }
    LaunchTimer lt(c, s, 1);
    LAUNCH(encode_blocks_kernel, dim3(grid), dim3(64 * kEncWaves),
           dbg_env("SNAPPY_HIP_ENC_LDS") ? atoi(dbg_env("SNAPPY_HIP_ENC_LDS")) : 0 /* DEBUG: fewer blocks per CU */, s, p);
  }
  if (dbg_env("SNAPPY_HIP_ENC_DBG")) {
    uint32_t q[4];
    HIP_TRY(hipMemcpyAsync(q, p.queue, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    fprintf(stderr, "ENC blocks taken by table-in-LDS waves %u, by table-in-memory waves %u (g_per4 %u)\n", q[1], q[2], p.g_per4);
  }
  if (d_estats) {
    unsigned long long h[16];
    HIP_TRY(hipMemcpyAsync(h, d_estats, 128, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const double r = h[8] ? (double)h[8] : 1.0;
    fprintf(stderr, "ENC STATS per block: fast rounds %.1f, left the fast loop: nothing from the first end %.1f, order %.2f; continuing rounds %.1f, fresh without a copy before %.1f\n",
            h[9] / (double)nb, h[10] / (double)nb, h[11] / (double)nb, h[12] / (double)nb, h[13] / (double)nb);
    fprintf(stderr, "ENC STATS rounds/block %.0f; ticks per round: probe %.0f table+fetch %.0f drain %.0f wait+compare %.0f "
            "chain-pre %.0f chain-hops %.0f chain-post %.0f repair %.0f\n", r / nb, h[0] / r, h[1] / r, h[2] / r, h[3] / r,
            h[4] / r, h[5] / r, h[6] / r, h[7] / r);
    (void)hipFree(d_estats);
  }
  HIP_TRY(hipGetLastError());
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_pack_d(snappy_hip_ctx* c, const uint8_t* d_slots, uint32_t slot_stride,
                                 const uint32_t* d_sizes, uint64_t n_blocks, uint64_t base,
                                 uint8_t* d_out, uint64_t* d_offsets, void* stream) {
  DeviceGuard guard(c->device);
  hipStream_t s = pick_stream(c, stream);
  StreamTurn turn(c, s);
  LaunchTimer lt(c, s, 3);
  LAUNCH(scan_sizes_kernel, dim3(1), dim3(kScanThreads), 0, s, d_sizes, n_blocks, base,
                     d_offsets);
  if (n_blocks)
    LAUNCH(gather_slots_kernel, dim3((uint32_t)n_blocks), dim3(256), 0, s, d_slots,
                       slot_stride, d_sizes, d_offsets, n_blocks, d_out);
  HIP_TRY(hipGetLastError());
  return SNAPPY_HIP_OK;
}

namespace {
constexpr uint64_t kIndexTeamMaxUnits = 1024;  // (four workgroups of four waves a CU: 1 024 units are resident at once)
int decode_d(snappy_hip_ctx* c, const uint8_t* d_in, const uint64_t* d_in_off,
             const uint32_t* d_in_len, uint64_t n_units, int unit, const uint8_t* d_kind,
             uint8_t* d_out, const uint64_t* d_out_off, const uint32_t* d_out_cap,
             uint32_t* d_out_len, uint32_t* d_status, bool stream_pass, hipStream_t s,
             uint32_t* d_crc = nullptr, const std::function<int()>* beside_index = nullptr) {
  // (beside_index: work the caller wants launched on another stream right in front of the index pass, the first
  // launch here that fills the GPU -- launched earlier, its workgroups hold up the small launches in front of it)
  if (n_units == 0) return beside_index ? (*beside_index)() : SNAPPY_HIP_OK;
  if (n_units > 0x7fffffffull) return SNAPPY_HIP_INVALID_INPUT;
  DecodeParams p{};
  p.in = d_in;
  p.in_off = d_in_off;
  p.in_len = d_in_len;
  p.out = d_out;
  p.out_off = d_out_off;
  p.out_cap = d_out_cap;
  p.out_len = d_out_len;
  p.status = d_status;
  p.kind = d_kind;
  p.n_units = n_units;
  p.unit = unit;
  if (const char* e = dbg_env("SNAPPY_HIP_DBG")) p.dbg = atoi(e);
  const bool v1 = d_kind != nullptr || dbg_env("SNAPPY_HIP_DECODE_V1") != nullptr;
  void* d_done = nullptr;  // per unit: the indexed decode kernel has written its CRC
  if (v1) {
    if (beside_index) {
      const int hs = (*beside_index)();
      if (hs) return hs;
    }
    LaunchTimer lt(c, s, 0);
    LAUNCH(decode_units_kernel<false>, dim3((uint32_t)n_units), dim3(64), 0, s, p);
  } else {
    // v2: index pass (where do elements start) + indexed block decode
    const uint64_t stride = kMaxRegionsPerUnit;
    void *d_idx, *d_cnt = nullptr, *d_ioff = nullptr;
    int st;
    if (n_units * stride * 4 <= (4ull << 30)) {
      if ((st = ws_get(c, 13, n_units * stride * 4, &d_idx))) return st;
    } else {
      // many units: compact index, sized by a scan of the per-unit region counts
      if ((st = ws_get(c, 11, n_units * 4, &d_cnt))) return st;
      if ((st = ws_get(c, 12, (n_units + 1) * 8, &d_ioff))) return st;
      LAUNCH(region_counts_kernel, dim3((uint32_t)((n_units + 255) / 256)), dim3(256),
                         0, s, d_in_len, n_units, (uint32_t*)d_cnt);
      LAUNCH(scan_sizes_kernel, dim3(1), dim3(kScanThreads), 0, s,
                         (const uint32_t*)d_cnt, n_units, (uint64_t)0, (uint64_t*)d_ioff);
      uint64_t total = 0;
      HIP_TRY(hipMemcpyAsync(&total, (uint64_t*)d_ioff + n_units, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if ((st = ws_get(c, 13, total * 4 + 64, &d_idx))) return st;
    }
    IndexParams ip{};
    ip.in = d_in;
    ip.in_off = d_in_off;
    ip.in_len = d_in_len;
    ip.out_cap = d_out_cap;
    ip.out_len = d_out_len;
    ip.status = d_status;
    ip.idx_off = (const uint64_t*)d_ioff;
    ip.idx_stride = stride;
    ip.idx = (uint32_t*)d_idx;
    ip.n_units = n_units;
    ip.unit = unit;
    Decode2Params dp{};
    dp.timeouts = c->d_counters;
    dp.in = d_in;
    dp.in_off = d_in_off;
    dp.in_len = d_in_len;
    dp.out = d_out;
    dp.out_off = d_out_off;
    dp.out_len = d_out_len;
    dp.status = d_status;
    dp.idx_off = (const uint64_t*)d_ioff;
    dp.idx_stride = stride;
    dp.idx = (const uint32_t*)d_idx;
    dp.n_units = n_units;
    dp.unit = unit;
    dp.tag_lut = c->d_tag_lut;
    if (const char* e = dbg_env("SNAPPY_HIP_DBG")) dp.dbg = atoi(e);
    // launch order: similar lengths together, longest first (1 024 units are resident all at once -- four ring workgroups
    // a CU: no order to choose, and five small launches less in front of a small call)
    if (n_units > 1024 && c->launch_order && !dbg_env("SNAPPY_HIP_NO_ORDER")) {
      void* d_perm;
      if ((st = launch_order(c, d_in_len, n_units, kOrderByLength, 15, 0, s, &d_perm))) return st;
      dp.order = (const uint32_t*)d_perm;
      ip.order = (const uint32_t*)d_perm;
      if (!dbg_env("SNAPPY_HIP_NO_SPREAD")) {  // the index pass: the sorted list walked with a golden-ratio stride (crc_pack_kernels.h)
        void* d_spread;
        if ((st = ws_get(c, 22, n_units * 4, &d_spread))) return st;
        uint64_t stride = (uint64_t)((double)n_units * 0.6180339887498949) | 1;
        auto gcd = [](uint64_t a, uint64_t b) {
          while (b) {
            const uint64_t t = a % b;
            a = b;
            b = t;
          }
          return a;
        };
        while (gcd(stride, n_units) != 1) stride += 2;
        LAUNCH(order_spread_kernel, dim3((uint32_t)((n_units + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_perm, n_units,
               stride % n_units, (uint32_t*)d_spread);
        ip.order = (const uint32_t*)d_spread;
      }
    }
    if (d_crc && kD2FusedCrc && !dbg_env("SNAPPY_HIP_NO_FUSED_CRC")) {  // the CRC comes out of the decode kernel's flush
      if ((st = ws_get(c, 14, n_units, &d_done))) return st;
      HIP_TRY(hipMemsetAsync(d_done, 0, n_units, s));
      dp.crc = d_crc;
      dp.crc_done = (uint8_t*)d_done;
      dp.crc_tab = c->d_crc_tab;
      dp.crc_col = c->d_col_mul;
      dp.crc_k32k = c->crc_k32k;
    }
    unsigned long long* d_stats = nullptr;
    if (dbg_env("SNAPPY_HIP_STATS")) {  // DEBUG
      HIP_TRY(hipMalloc((void**)&d_stats, 256));
      HIP_TRY(hipMemsetAsync(d_stats, 0, 256, s));
      dp.stats = d_stats;
    }
    // (units of few, long elements are decoded by the index pass's own waves, sparse_kernel.h)
    const bool sparse_on = kD2RingFirst && !dbg_env("SNAPPY_HIP_NO_RING") && !dbg_env("SNAPPY_HIP_NO_SPARSE") && !dbg_env("SNAPPY_HIP_NO_ONEPASS");
    ip.sparse = sparse_on ? 1 : 0;
    ip.out = d_out;
    ip.out_off = d_out_off;
    ip.sparse_counters = c->d_counters + 2;
    if (beside_index && (st = (*beside_index)())) return st;
    beside_index = nullptr;
    {
      LaunchTimer lt(c, s, 4);
      // a small batch: four waves a unit (index_kernel.h, TEAM) -- its units leave most of the GPU's wave slots empty
      // anyway, and a unit's walk by one wave is ~250 us whatever the batch
      if (n_units <= kIndexTeamMaxUnits && !dbg_env("SNAPPY_HIP_NO_INDEX_TEAM"))
        LAUNCH((index_units_kernel<false, true>), dim3((uint32_t)n_units), dim3(64 * kSplitWaves), 0, s, ip);
      else
        LAUNCH(index_units_kernel<false>, dim3((uint32_t)n_units), dim3(64), 0, s, ip);
    }
    if (dbg_env("SNAPPY_HIP_VERIFY_INDEX")) {  // DEBUG
      uint32_t* d_rep;
      HIP_TRY(hipMalloc((void**)&d_rep, n_units * 16));
      LAUNCH(verify_index_kernel, dim3((uint32_t)((n_units + 63) / 64)), dim3(64), 0, s, ip, d_rep);
      std::vector<uint32_t> rep(n_units * 4);
      HIP_TRY(hipMemcpyAsync(rep.data(), d_rep, n_units * 16, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      int shown = 0;
      for (uint64_t i = 0; i < n_units; i++)
        if (rep[i * 4] != 0xffffffffu && shown++ < 12)
          fprintf(stderr, "INDEX MISMATCH unit %llu region %u (chunk %u lane %u): want off %u nc %u dst %u, got off %u nc %u dst %u\n",
                  (unsigned long long)i, rep[i * 4], rep[i * 4] / 64, rep[i * 4] % 64, rep[i * 4 + 1] & 63,
                  (rep[i * 4 + 1] >> 6) & 31, rep[i * 4 + 1] >> 11, rep[i * 4 + 2] & 63,
                  (rep[i * 4 + 2] >> 6) & 31, rep[i * 4 + 2] >> 11);
      fprintf(stderr, "index verify: %d units with a mismatch\n", shown);
      (void)hipFree(d_rep);
    }
    // The ring-window instantiation first, four workgroups per CU (with the CRC wanted: the variant that
    // checksums the rows its flush completes); then the whole-block one over the units it passed on (a
    // workgroup of any other unit leaves at once).  kD2RingFirst == 0 / SNAPPY_HIP_NO_RING (debug builds):
    // whole-block only.
    const bool ring_first = kD2RingFirst && !dbg_env("SNAPPY_HIP_NO_RING");
    if (ring_first) {
      LaunchTimer lt(c, s, 0);
      bool launched = false;
      if constexpr (kD2FusedCrc) {
        if (dp.crc) {  // (the CRC out of the ring's flush: the framed stream's chunks, snappy.nim:231)
          LAUNCH((decode_indexed_kernel<kRingWin, true>), dim3((uint32_t)n_units), dim3(kD2Threads), 0, s, dp);
          launched = true;
        }
      }
      if (!launched) LAUNCH(decode_indexed_kernel<kRingWin>, dim3((uint32_t)n_units), dim3(kD2Threads), 0, s, dp);  // (static window)
    }
    if (ring_first) {  // the units it passed on, as a list
      void* d_pass;
      if ((st = ws_get(c, 20, 8 + n_units * 4, &d_pass))) return st;
      HIP_TRY(hipMemsetAsync(d_pass, 0, 8, s));
      dp.pass_list = (const uint32_t*)d_pass + 2;
      LAUNCH(passed_on_list_kernel, dim3((uint32_t)((n_units + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_status,
             dp.order, n_units, (uint32_t*)d_pass + 2);
    }
    {
      LaunchTimer lt(c, s, ring_first ? 8 : 0);
      dp.second = ring_first ? 1 : 0;
      LAUNCH(decode_indexed_kernel<kMaxBlockLen>, dim3((uint32_t)n_units), dim3(kD2Threads),
             kD2DynWindow + ((dbg_env("SNAPPY_HIP_ONE_WG") || kD2Threads > 640) ? 8192 : 0) /* one per CU */, s, dp);
    }
    if (d_stats) {
      unsigned long long h[32];
      HIP_TRY(hipMemcpyAsync(h, d_stats, 256, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      for (int w = 0; w < 2; w++) {  // waves 0 and 5 of every workgroup, summed
        const unsigned long long* q = h + 12 * w;
        const double st = q[1] ? (double)q[1] : 1.0;  // steps
        fprintf(stderr,
                "STATS wave %d: steps %llu; per step: groups %.2f doubling rounds/group %.2f | ticks: flush %.0f front end %.0f "
                "prep %.0f wait %.0f turn %.0f barrier %.0f (work %.0f)\n",
                w ? 5 : 0, q[1], q[2] / st, q[2] ? (double)q[3] / q[2] : 0.0, q[6] / st, q[7] / st, q[8] / st, q[9] / st,
                q[10] / st, q[5] / st, q[4] / st);
      }
      (void)hipFree(d_stats);
    }
  }
  if (!v1 && !dbg_env("SNAPPY_HIP_NO_ONEPASS")) {  // units the indexed decoder declined: listed, then decoded
    void* d_list;
    int st = ws_get(c, 23, 8 + n_units * 4, &d_list);
    if (st) return st;
    HIP_TRY(hipMemsetAsync(d_list, 0, 8, s));
    LaunchTimer lt(c, s, 5);
    LAUNCH(decode_finish_kernel, dim3((uint32_t)((n_units + 255) / 256)), dim3(256), 0, s, d_status, n_units, (uint32_t*)d_list + 2);
    p.only_status = kNeedsOnePass;
    p.list = (const uint32_t*)d_list + 2;
    LAUNCH(decode_units_kernel<false>, dim3((uint32_t)(n_units < 512 ? n_units : 512)), dim3(64), 0, s, p);
    p.only_status = 0;
    p.list = nullptr;
  }
  if (stream_pass) {
    LaunchTimer lt(c, s, 5);
    LAUNCH(decode_units_kernel<true>, dim3((uint32_t)n_units), dim3(64), 0, s, p);
  }
  if (d_crc) {  // units that did not get their CRC from the indexed decode kernel (all of them for v1)
    CrcParams cp{};
    cp.in = d_out;
    cp.off = d_out_off;
    cp.len = d_out_len;
    cp.crc = d_crc;
    cp.n_units = n_units;
    cp.stride_tab = c->d_crc_tab;
    cp.col_mul = c->d_col_mul;
    cp.done = (const uint8_t*)d_done;
    LaunchTimer lt(c, s, 2);
    LAUNCH(crc32c_units_kernel, dim3((uint32_t)n_units), dim3(kCrcThreads), 0, s, cp);
  }
  HIP_TRY(hipGetLastError());
  return SNAPPY_HIP_OK;
}
}  // namespace

extern "C" int snappy_hip_decode_blocks_d(snappy_hip_ctx* c, const uint8_t* d_in,
                                          const uint64_t* d_in_off, const uint32_t* d_in_len,
                                          uint64_t n_units, int unit, uint8_t* d_out,
                                          const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                          uint32_t* d_out_len, uint32_t* d_status, uint32_t* d_crc,
                                          void* stream) {
  if (unit != kUnitBody && unit != kUnitRaw) return SNAPPY_HIP_INVALID_INPUT;
  DeviceGuard guard(c->device);
  hipStream_t s = pick_stream(c, stream);
  StreamTurn turn(c, s);
  return decode_d(c, d_in, d_in_off, d_in_len, n_units, unit, nullptr, d_out, d_out_off, d_out_cap,
                  d_out_len, d_status, true, s, d_crc);
}

// compressFramed (snappy.nim:130-155) for an input that is resident in HBM: the stream identifier,
// then one chunk per 65 536-byte slice (encodeFrame, encoder.nim:385-426, in the encode kernel),
// packed contiguously.  *written (host) receives the stream's length; the call returns when the
// stream is complete (it reads the scanned total back).
extern "C" int snappy_hip_compress_framed_d(snappy_hip_ctx* c, const uint8_t* d_in, uint64_t n,
                                            uint8_t* d_out, uint64_t cap, uint64_t* written,
                                            void* stream) {
  *written = 0;
  if (cap < snappy_hip_max_compressed_len_framed((int64_t)n)) return SNAPPY_HIP_BUFFER_TOO_SMALL;  // snappy.nim:139-140
  DeviceGuard guard(c->device);
  hipStream_t s = pick_stream(c, stream);
  StreamTurn turn(c, s);
  HIP_TRY(hipMemcpyAsync(d_out, kFramingHeader, sizeof kFramingHeader, hipMemcpyHostToDevice, s));
  const uint64_t nb = (n + kMaxBlockLen - 1) / kMaxBlockLen;
  uint64_t end = sizeof kFramingHeader;
  if (nb) {
    void *d_slots, *d_sizes, *d_offsets;
    int st;
    if ((st = ws_get(c, 17, nb * (size_t)kSlotStride, &d_slots))) return st;
    if ((st = ws_get(c, 18, nb * 4 + (nb + 1) * 8 + 64, &d_sizes))) return st;
    d_offsets = (uint8_t*)d_sizes + ((nb * 4 + 15) & ~(size_t)15);
    if ((st = snappy_hip_encode_blocks_d(c, d_in, n, kMaxBlockLen, kUnitFrame, (uint8_t*)d_slots, kSlotStride,
                                         (uint32_t*)d_sizes, s)))
      return st;
    if ((st = snappy_hip_pack_d(c, (const uint8_t*)d_slots, kSlotStride, (const uint32_t*)d_sizes, nb,
                                sizeof kFramingHeader, d_out, (uint64_t*)d_offsets, s)))
      return st;
    HIP_TRY(hipMemcpyAsync(&end, (uint64_t*)d_offsets + nb, 8, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  *written = end;
  return SNAPPY_HIP_OK;
}

// uncompressFramed (snappy.nim:169-267) for a stream that is resident in HBM: chunk walk, block
// decode, CRC comparison and the first-failure verdict all run on the device (framed_kernels.h);
// what comes back to the host is the verdict: status and the two counters.
extern "C" int snappy_hip_uncompress_framed_d(snappy_hip_ctx* c, const uint8_t* d_in, uint64_t n,
                                              uint8_t* d_out, uint64_t cap, int check_header,
                                              int check_integrity, uint64_t* read_out,
                                              uint64_t* written_out, void* stream) {
  *read_out = 0;
  *written_out = 0;
  DeviceGuard guard(c->device);
  hipStream_t s = pick_stream(c, stream);
  StreamTurn turn(c, s);
  FrameScanResult res{};
  FrameUnits comp{}, stored{};
  uint32_t *comp_status = nullptr, *comp_len = nullptr, *comp_crc = nullptr, *stored_crc = nullptr;
  FrameScanResult* d_res = nullptr;
  FrameVerdict* d_verdict = nullptr;
  auto carve_lists = [&](size_t cap_l) -> int {
    const size_t per_list = cap_l * (8 + 4 + 8 + 4 + 4 + 4 + 8);
    void* base;
    int st = ws_get(c, 19, 2 * per_list + cap_l * 16 + 256, &base);
    if (st) return st;
    uint8_t* q = (uint8_t*)base;
    auto carve = [&](FrameUnits* u) {
      u->in_off = (uint64_t*)q, q += cap_l * 8;
      u->out_off = (uint64_t*)q, q += cap_l * 8;
      u->hdr_at = (uint64_t*)q, q += cap_l * 8;
      u->in_len = (uint32_t*)q, q += cap_l * 4;
      u->out_cap = (uint32_t*)q, q += cap_l * 4;
      u->crc = (uint32_t*)q, q += cap_l * 4;
      u->seq = (uint32_t*)q, q += cap_l * 4;
    };
    carve(&comp);
    carve(&stored);
    comp_status = (uint32_t*)q, q += cap_l * 4;
    comp_len = (uint32_t*)q, q += cap_l * 4;
    comp_crc = (uint32_t*)q, q += cap_l * 4;
    stored_crc = (uint32_t*)q, q += cap_l * 4;
    d_res = (FrameScanResult*)q, q += 128;
    d_verdict = (FrameVerdict*)q;
    return SNAPPY_HIP_OK;
  };
  bool have_lists = false;
  // ---- the chunk walk in parallel (framed_kernels.h): well-formed streams of at least a few MiB ----
  if (n >= (4u << 20) && n <= 0xffffffffull * 256) {
    const uint64_t p0 = check_header ? sizeof kFramingHeader : 0;
    // (slices of at least 1 MiB: a chaser then has a few dozen chunks to chase, and a small stream few chasers)
    uint64_t slice = ((n + kFrameChasers - 1) / kFrameChasers + 15) & ~15ull;
    if (slice < (1u << 20)) slice = 1u << 20;
    void* wbase;
    int st = ws_get(c, 15, (size_t)kFrameChasers * (sizeof(FrameChase) + kChaserList * 8 + 8) + 256, &wbase);
    if (st) return st;
    uint8_t* q = (uint8_t*)wbase;
    uint64_t* d_lists = (uint64_t*)q;
    q += (size_t)kFrameChasers * kChaserList * 8;
    FrameChase* d_chase = (FrameChase*)q;
    q += (size_t)kFrameChasers * sizeof(FrameChase);
    uint32_t* d_base = (uint32_t*)q;
    q += ((size_t)kFrameChasers + 1) * 4;
    uint32_t* d_first = (uint32_t*)q;
    q += (size_t)kFrameChasers * 4;
    q = (uint8_t*)(((uintptr_t)q + 15) & ~(uintptr_t)15);
    FrameStitch* d_stitch = (FrameStitch*)q;
    FrameStitch stitch{};
    {
      LaunchTimer lt(c, s, 6);
      LAUNCH(frame_chase_kernel, dim3(kFrameChasers), dim3(64), 0, s, d_in, n, p0, slice, d_chase, d_lists);
      LAUNCH(frame_stitch_kernel, dim3(1), dim3(1024), 0, s, d_chase, d_lists, n, p0, slice, d_base, d_first,
                         d_stitch);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&stitch, d_stitch, sizeof stitch, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (dbg_env("SNAPPY_HIP_STATS"))  // DEBUG
      fprintf(stderr, "FRAME WALK stitch irregular %u chunks %u slice %llu\n", stitch.irregular, stitch.n_chunks,
              (unsigned long long)slice);
    if (!stitch.irregular && stitch.n_chunks) {
      const size_t H = stitch.n_chunks;
      if ((st = carve_lists(H + 16))) return st;
      void* fbase;
      if ((st = ws_get(c, 14, H * (8 + 4 + 4 + 4) + 3 * (H + 1) * 8 + 256, &fbase))) return st;
      uint8_t* f = (uint8_t*)fbase;
      uint64_t* d_pos = (uint64_t*)f;
      f += H * 8;
      uint64_t* d_out_at = (uint64_t*)f;
      f += (H + 1) * 8;
      uint64_t* d_comp_at = (uint64_t*)f;
      f += (H + 1) * 8;
      uint64_t* d_stored_at = (uint64_t*)f;
      f += (H + 1) * 8;
      uint32_t* d_ulen = (uint32_t*)f;
      f += H * 4;
      uint32_t* d_is_comp = (uint32_t*)f;
      f += H * 4;
      uint32_t* d_is_stored = (uint32_t*)f;
      f += H * 4;
      uint32_t* d_flags = (uint32_t*)f;  // [0] irregular, [1] fast_ok
      HIP_TRY(hipMemsetAsync(d_flags, 0, 8, s));
      FrameFillParams fp{};
      fp.in = d_in;
      fp.n = n;
      fp.lists = d_lists;
      fp.base = d_base;
      fp.first = d_first;
      fp.stitch = d_stitch;
      fp.pos = d_pos;
      fp.ulen = d_ulen;
      fp.is_comp = d_is_comp;
      fp.is_stored = d_is_stored;
      fp.irregular = d_flags;
      const uint32_t grid = (uint32_t)((H + 255) / 256);
      LAUNCH(frame_fill_kernel, dim3(grid), dim3(256), 0, s, fp);
      {  // the three scans side by side, a workgroup each
        ScanJobs sj{};
        sj.sizes[0] = d_ulen, sj.offsets[0] = d_out_at;
        sj.sizes[1] = d_is_comp, sj.offsets[1] = d_comp_at;
        sj.sizes[2] = d_is_stored, sj.offsets[2] = d_stored_at;
        sj.n = (uint64_t)H;
        // (tiles, a workgroup each, up to 2^20 chunks; beyond that a tile's own sum of what lies in front of it would be
        // the longer part: one workgroup per scan, pass by pass)
        const uint32_t tiles = H <= (1u << 20) ? (uint32_t)((H + kScanTile - 1) / kScanTile) : 1u;
        LAUNCH(scan_sizes_jobs_kernel, dim3(3, tiles ? tiles : 1u), dim3(kScanThreads), 0, s, sj);
      }
      FrameScatterParams xp{};
      xp.in = d_in;
      xp.n = n;
      xp.cap = cap;
      xp.stitch = d_stitch;
      xp.irregular = d_flags;
      xp.pos = d_pos;
      xp.ulen = d_ulen;
      xp.is_comp = d_is_comp;
      xp.is_stored = d_is_stored;
      xp.out_at = d_out_at;
      xp.comp_at = d_comp_at;
      xp.stored_at = d_stored_at;
      xp.comp = comp;
      xp.stored = stored;
      xp.list_cap = (uint32_t)(H + 16);
      xp.check_header = check_header;
      xp.res = d_res;
      xp.fast_ok = d_flags + 1;
      LAUNCH(frame_scatter_kernel, dim3(grid), dim3(256), 0, s, xp);
      HIP_TRY(hipGetLastError());
      uint32_t flags[2] = {0, 0};
      HIP_TRY(hipMemcpyAsync(flags, d_flags, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(&res, d_res, sizeof res, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      have_lists = flags[1] == 1;
      if (dbg_env("SNAPPY_HIP_STATS")) fprintf(stderr, "FRAME WALK fill irregular %u fast_ok %u\n", flags[0], flags[1]);
    }
  }
  uint64_t need_lists = 0;
  for (int attempt = 0; !have_lists; attempt++) {  // the serial walk: every stream, every verdict
    // chunk lists: room for one chunk per KiB of stream at first; a stream of tinier chunks makes the walk
    // count on behind the full list, and the second attempt's lists are sized by that count
    const uint64_t want = attempt == 0 ? n / 1024 + 4096 : need_lists + 16;
    if (want > 0x7fffffffull) {  // (more than 2^31 chunks: a device limit, not a malformed stream)
      g_last_error = "framed stream of more than 2^31 chunks";
      return SNAPPY_HIP_DEVICE_ERROR;
    }
    const size_t cap_l = (size_t)want;
    int st = carve_lists(cap_l);
    if (st) return st;
    FrameScanParams sp{};
    sp.in = d_in;
    sp.n = n;
    sp.cap = cap;
    sp.check_header = check_header;
    sp.check_integrity = check_integrity;
    sp.comp = comp;
    sp.stored = stored;
    sp.list_cap = (uint32_t)cap_l;
    sp.res = d_res;
    {
      LaunchTimer lt(c, s, 6);
      LAUNCH(frame_scan_kernel, dim3(1), dim3(64), 0, s, sp);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&res, d_res, sizeof res, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (!res.overflow) break;
    need_lists = res.need_comp > res.need_stored ? res.need_comp : res.need_stored;
    if (attempt == 1) {
      g_last_error = "internal: chunk lists overflowed twice";
      return SNAPPY_HIP_DEVICE_ERROR;
    }
  }
  // The stored chunks are checksummed (snappy.nim:244) and copied (:256) in one pass over their bytes -- on a second
  // stream, beside the compressed chunks' decode: the two touch different bytes, and the decode's index pass is bound
  // by the vector ALU, not by memory.  (The lists are complete: the stream was waited for above.)
  const std::function<int()> launch_stored = [&]() -> int {
    if (!res.n_stored) return SNAPPY_HIP_OK;
    hipStream_t s2 = c->side_stream;
    if (check_integrity) {
      CrcParams cp{};
      cp.in = d_in;
      cp.off = stored.in_off;
      cp.len = stored.in_len;
      cp.crc = stored_crc;
      cp.n_units = res.n_stored;
      cp.stride_tab = c->d_crc_tab;
      cp.col_mul = c->d_col_mul;
      cp.copy_out = d_out;
      cp.copy_off = stored.out_off;
      cp.copy_cap = stored.out_cap;  // (0 for a chunk that is only checksummed)
      LaunchTimer lt(c, s2, 2);
      LAUNCH(crc32c_units_kernel, dim3(res.n_stored), dim3(kCrcThreads), 0, s2, cp);
    } else {
      LAUNCH(copy_units_kernel, dim3(res.n_stored), dim3(256), 0, s2, d_in, stored.in_off, stored.out_cap,
                         stored.out_off, d_res, d_out);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->side_done, s2));
    return SNAPPY_HIP_OK;
  };
  if (!res.n_comp) {
    const int st = launch_stored();
    if (st) return st;
  } else {  // compressed chunks: uncompress() each (snappy.nim:216), checksummed from the decoder's window
    int st = decode_d(c, d_in, comp.in_off, comp.in_len, res.n_comp, kUnitRaw, nullptr, d_out, comp.out_off,
                      comp.out_cap, comp_len, comp_status, true, s, check_integrity ? comp_crc : nullptr, &launch_stored);
    if (st) {
      if (res.n_stored) (void)hipStreamSynchronize(c->side_stream);
      return st;
    }
  }
  if (res.n_stored) HIP_TRY(hipStreamWaitEvent(s, c->side_done, 0));
  FrameVerdictParams vp{};
  vp.comp = comp;
  vp.stored = stored;
  vp.res = d_res;
  vp.comp_status = comp_status;
  vp.comp_crc = comp_crc;
  vp.stored_crc = stored_crc;
  vp.check_integrity = check_integrity;
  vp.out = d_verdict;
  LAUNCH(frame_verdict_kernel, dim3(1), dim3(1024), 0, s, vp);
  HIP_TRY(hipGetLastError());
  FrameVerdict v{};
  HIP_TRY(hipMemcpyAsync(&v, d_verdict, sizeof v, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (v.status != kOk) return (int)v.status;
  *read_out = v.read;
  *written_out = v.written;
  return SNAPPY_HIP_OK;
}

// Block-range sharding over several contexts (one per GPU) from ONE process, with the host-side
// concatenate of BASELINE's north star: shard k (a contiguous range of 64 KiB blocks, resident on
// context k's GPU) is encoded and packed there; the only exchange is the n shard totals, whose
// exclusive scan -- the reference's serial `written += ...`, snappy.nim:56-62, :149-153 -- gives every
// shard its place in the ONE host buffer, and all shards then download side by side, each over its
// own GPU's link.  framed != 0: compressFramed (stream identifier + one chunk per block, any total
// length); framed == 0: compress (one varint, total < 2^32).  out should be page-locked for the
// downloads to overlap.  shard_off (may be NULL) receives the n + 1 scanned offsets.
// Block ranges on n contexts, host-side concatenate (snappy.nim:56-62, :146-153), in STAGES so that a GPU encodes
// while its earlier output travels: input k holds, back to back, the blocks of context k's stages; stage j of context
// k is the global block range [(j n + k) S, (j n + k + 1) S) (S = stage_blocks; the last stage may be short, its ranges
// still in context order), so the stream is the stages in order, each the contexts in order.  A context's thread
// encodes + packs its stage, publishes the stage's size, and downloads -- on a second stream, behind an event -- every
// earlier stage of its own whose place in `out` has become known (the sum of everything in front of it: the one
// exchange of this path, n integers per stage through host memory), then goes on with the next stage: nobody waits
// for a neighbour before its last stage.  stage_blocks = 0: one stage, i.e. n contiguous shards.
extern "C" int snappy_hip_compress_shards_staged(snappy_hip_ctx* const* ctxs, int n, const uint8_t* const* d_in,
                                                 const uint64_t* in_len, uint64_t stage_blocks, int framed, uint8_t* out,
                                                 uint64_t cap, uint64_t* written, uint64_t* shard_off) {
  *written = 0;
  if (n <= 0) return SNAPPY_HIP_INVALID_INPUT;
  for (int k = 0; k < n; k++) {  // a context serves one shard at a time (it owns the scratch buffers its calls use)
    if (!ctxs[k]) return SNAPPY_HIP_INVALID_INPUT;
    for (int j = 0; j < k; j++)
      if (ctxs[j] == ctxs[k]) return SNAPPY_HIP_INVALID_INPUT;
  }
  uint64_t total_in = 0;
  for (int k = 0; k < n; k++) total_in += in_len[k];
  const uint64_t total_blocks = (total_in + kMaxBlockLen - 1) / kMaxBlockLen;
  // (a stage never needs to be larger than the input: keeps S * 65536 and S * n far from wrapping)
  if (stage_blocks > total_blocks) stage_blocks = total_blocks ? total_blocks : 1;
  const uint64_t S = stage_blocks ? stage_blocks : ((uint64_t)1 << 40);
  const uint64_t per_stage = S * (uint64_t)n;  // blocks of a full stage
  const uint64_t n_stages = stage_blocks ? (total_blocks + per_stage - 1) / per_stage : 1;
  // what context k holds of stage j, in bytes, by the layout above -- and the caller's lengths must agree with it
  auto stage_bytes = [&](uint64_t j, int k) -> uint64_t {
    if (!stage_blocks) return in_len[k];
    const uint64_t lo = (j * n + (uint64_t)k) * S * kMaxBlockLen, hi = lo + S * kMaxBlockLen;
    return total_in <= lo ? 0 : (total_in < hi ? total_in - lo : hi - lo);
  };
  for (int k = 0; k < n; k++) {
    uint64_t sum = 0;
    for (uint64_t j = 0; j < n_stages; j++) sum += stage_bytes(j, k);
    if (sum != in_len[k]) return SNAPPY_HIP_INVALID_INPUT;
    if (!stage_blocks && k + 1 < n && in_len[k] % kMaxBlockLen) return SNAPPY_HIP_INVALID_INPUT;  // shards are whole blocks
  }
  if (!framed && total_in > 0xffffffffull) return SNAPPY_HIP_INVALID_INPUT;  // snappy.nim:41-42
  const uint64_t need = framed ? snappy_hip_max_compressed_len_framed((int64_t)total_in)
                               : 32 + total_in + total_in / 6;
  if (cap < need) return SNAPPY_HIP_BUFFER_TOO_SMALL;  // snappy.nim:44-45, :139-140
  uint64_t base = 0;
  if (framed) {
    memcpy(out, kFramingHeader, sizeof kFramingHeader);
    base = sizeof kFramingHeader;
  } else {
    base = (uint64_t)varint_encode_u32((uint32_t)total_in, out);
  }
  // sizes[j * n + k]: bytes of stage j of context k in the stream (~0: not known yet); the stream order is this index
  std::vector<uint64_t> sizes(n_stages * n, ~0ull);
  std::mutex mu;
  std::condition_variable cv;
  bool failed = false;
  int first_status = SNAPPY_HIP_OK;  // the FIRST failure (the others then fail because of it): what the caller is told
  std::string first_err;
  std::vector<int> status(n, SNAPPY_HIP_OK);
  std::vector<std::string> errs(n);
  const int unit = framed ? kUnitFrame : kUnitBody;
  // offset of stream piece `idx` if everything in front of it is known, else ~0 (caller holds mu)
  auto place_of = [&](uint64_t idx) -> uint64_t {
    uint64_t at = base;
    for (uint64_t i = 0; i < idx; i++) {
      if (sizes[i] == ~0ull) return ~0ull;
      at += sizes[i];
    }
    return at;
  };
  auto worker = [&](int k) {
    snappy_hip_ctx* c = ctxs[k];
    DeviceGuard guard(c->device);
    auto fail = [&](int st) {
      status[k] = st;
      errs[k] = g_last_error;
      std::lock_guard<std::mutex> lk(mu);
      if (!failed) {
        first_status = st;
        first_err = g_last_error;
      }
      failed = true;
      for (uint64_t j = 0; j < n_stages; j++)
        if (sizes[j * n + k] == ~0ull) sizes[j * n + k] = 0;  // (nobody waits for me)
      cv.notify_all();
    };
    const uint64_t nb_all = (in_len[k] + kMaxBlockLen - 1) / kMaxBlockLen;
    const uint64_t nb_stage = stage_blocks ? (S < nb_all ? S : nb_all) : nb_all;
    void *d_slots = nullptr, *d_sizes = nullptr, *d_out = nullptr;
    int st = 0;
    if (nb_all) {
      if ((st = ws_get(c, 17, nb_stage * (size_t)kSlotStride, &d_slots)) ||
          (st = ws_get(c, 18, nb_stage * 4 + (nb_stage + 1) * 8 + 64, &d_sizes)) ||
          (st = ws_get(c, 4, nb_all * (size_t)(kMaxCompressedBlockLen + 16) + 64, &d_out)))
        return fail(st);
    }
    hipStream_t copy_stream = nullptr;
    hipEvent_t packed_ev = nullptr;
    if (hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&packed_ev, hipEventDisableTiming) != hipSuccess) {
      if (copy_stream) (void)hipStreamDestroy(copy_stream);
      return fail(SNAPPY_HIP_DEVICE_ERROR);
    }
    void* d_offsets = d_sizes ? (uint8_t*)d_sizes + ((nb_stage * 4 + 15) & ~(size_t)15) : nullptr;
    std::vector<uint64_t> at_dev(n_stages + 1, 0);  // where my stages lie in d_out
    uint64_t in_at = 0, next_copy = 0;
    // download every stage of mine, in order, whose place is known (block = wait for it)
    auto copy_ready = [&](uint64_t upto_stage, bool block) -> bool {
      while (next_copy < upto_stage) {
        const uint64_t idx = next_copy * n + (uint64_t)k;
        uint64_t at;
        {
          std::unique_lock<std::mutex> lk(mu);
          at = place_of(idx);
          while (at == ~0ull && block && !failed) {
            cv.wait(lk);
            at = place_of(idx);
          }
          if (failed) return false;
        }
        if (at == ~0ull) return true;  // later
        const uint64_t len = at_dev[next_copy + 1] - at_dev[next_copy];
        if (at + len > cap) {
          g_last_error = "internal: packed stream exceeds the caller's bound";
          return false;
        }
        if (len && hipMemcpyAsync(out + at, (uint8_t*)d_out + at_dev[next_copy], len, hipMemcpyDeviceToHost, copy_stream) !=
                       hipSuccess)
          return false;
        next_copy++;
      }
      return true;
    };
    bool ok = true;
    for (uint64_t j = 0; j < n_stages && ok; j++) {
      const uint64_t bytes = stage_bytes(j, k);
      const uint64_t nb = (bytes + kMaxBlockLen - 1) / kMaxBlockLen;
      uint64_t end = at_dev[j];
      if (nb) {
        if ((st = snappy_hip_encode_blocks_d(c, d_in[k] + in_at, bytes, kMaxBlockLen, unit, (uint8_t*)d_slots, kSlotStride,
                                             (uint32_t*)d_sizes, nullptr)) ||
            (st = snappy_hip_pack_d(c, (const uint8_t*)d_slots, kSlotStride, (const uint32_t*)d_sizes, nb, at_dev[j],
                                    (uint8_t*)d_out, (uint64_t*)d_offsets, nullptr))) {
          (void)hipStreamSynchronize(copy_stream);  // (earlier stages' downloads into the caller's buffer may be in flight)
          (void)hipStreamDestroy(copy_stream);
          (void)hipEventDestroy(packed_ev);
          return fail(st);
        }
        if (hipMemcpyAsync(&end, (uint64_t*)d_offsets + nb, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipEventRecord(packed_ev, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
            hipStreamWaitEvent(copy_stream, packed_ev, 0) != hipSuccess)
          ok = false;
      }
      at_dev[j + 1] = end;
      in_at += bytes;
      {
        std::lock_guard<std::mutex> lk(mu);
        sizes[j * n + k] = end - at_dev[j];
        cv.notify_all();
      }
      ok = ok && copy_ready(j + 1, false);  // (what can go already goes; the next stage's kernels run beside it)
    }
    ok = ok && copy_ready(n_stages, true) && hipStreamSynchronize(copy_stream) == hipSuccess;
    (void)hipStreamSynchronize(copy_stream);
    (void)hipStreamDestroy(copy_stream);
    (void)hipEventDestroy(packed_ev);
    if (!ok) fail(SNAPPY_HIP_DEVICE_ERROR);
  };
  // (one thread per context; a context whose thread cannot be had would deadlock the others' waits for its sizes, so
  // such a failure is reported before anything runs)
  {
    std::vector<std::thread> th;
    try {
      for (int k = 1; k < n; k++) th.emplace_back(worker, k);
    } catch (...) {
      {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        for (auto& v : sizes)
          if (v == ~0ull) v = 0;
        cv.notify_all();
      }
      for (auto& t : th) t.join();
      g_last_error = "could not start a thread per context";
      return SNAPPY_HIP_DEVICE_ERROR;
    }
    worker(0);
    for (auto& t : th) t.join();
  }
  if (first_status) {
    g_last_error = first_err;
    return first_status;
  }
  for (int k = 0; k < n; k++)
    if (status[k]) {
      g_last_error = errs[k];
      return status[k];
    }
  uint64_t total = base;
  for (uint64_t v : sizes) total += v;
  if (shard_off) {  // where each context's FIRST stage lies, and the stream's end (contiguous shards: the n + 1 scanned offsets)
    uint64_t at = base;
    for (int k = 0; k < n; k++) {
      shard_off[k] = at;
      at += sizes[k];
    }
    shard_off[n] = total;
  }
  *written = total;
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_compress_shards(snappy_hip_ctx* const* ctxs, int n, const uint8_t* const* d_in,
                                          const uint64_t* in_len, int framed, uint8_t* out, uint64_t cap,
                                          uint64_t* written, uint64_t* shard_off) {
  return snappy_hip_compress_shards_staged(ctxs, n, d_in, in_len, 0, framed, out, cap, written, shard_off);
}

// =============================================================================================
// host-side scalar helpers
// =============================================================================================
extern "C" uint64_t snappy_hip_max_compressed_len(uint32_t n) {  // codec.nim:117-120
  return 32u + (uint64_t)n + (uint64_t)n / 6u;
}

extern "C" uint64_t snappy_hip_max_compressed_len_framed(int64_t n) {  // codec.nim:140-164
  if (n <= 0) return sizeof kFramingHeader;
  uint64_t frames = ((uint64_t)n + kMaxBlockLen - 1) / kMaxBlockLen;
  return (frames - 1) * (kMaxBlockLen + 8) + snappy_hip_max_compressed_len(kMaxBlockLen) + 8 +
         sizeof kFramingHeader;
}

extern "C" int snappy_hip_uncompressed_len(const uint8_t* in, size_t n, uint64_t* len) {
  uint64_t v;
  if (varint_decode(in, n, 64, &v) <= 0) return SNAPPY_HIP_INVALID_INPUT;  // codec.nim:134-138
  *len = v;
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_uncompressed_len_framed(const uint8_t* in, size_t n, uint64_t* len) {
  size_t rd = 0;  // codec.nim:178-214
  uint64_t expected = 0;
  while (rd < n) {
    size_t remaining = n - rd;
    if (remaining < 4) return SNAPPY_HIP_INVALID_INPUT;
    uint32_t hdr = (uint32_t)in[rd] | ((uint32_t)in[rd + 1] << 8) | ((uint32_t)in[rd + 2] << 16) |
                   ((uint32_t)in[rd + 3] << 24);
    uint8_t id = (uint8_t)hdr;
    size_t data_len = hdr >> 8;
    if (remaining < data_len + 4) return SNAPPY_HIP_INVALID_INPUT;
    rd += 4;
    uint64_t u = 0;
    if (id == 0x00) {
      if (data_len < 4) return SNAPPY_HIP_INVALID_INPUT;
      if (varint_decode(in + rd + 4, data_len - 4, 64, &u) <= 0) return SNAPPY_HIP_INVALID_INPUT;
    } else if (id == 0x01) {
      if (data_len < 4) return SNAPPY_HIP_INVALID_INPUT;
      u = data_len - 4;
    } else if (id < 0x80) {
      return SNAPPY_HIP_INVALID_INPUT;
    }
    if (u > kMaxBlockLen) return SNAPPY_HIP_INVALID_INPUT;
    expected += u;
    rd += data_len;
  }
  *len = expected;
  return SNAPPY_HIP_OK;
}

// =============================================================================================
// host-buffer API
// =============================================================================================
namespace {

// The host-buffer calls run on contexts from a pool: a call takes a free context (or makes one),
// and gives it back when it returns, so concurrent callers -- the reference's API is re-entrant --
// run side by side, each on its own stream and scratch buffers, and the number of contexts is the
// largest number of calls that ever ran at once.  (SNAPPY_HIP_DEVICE picks the GPU.)
std::mutex g_pool_mu;
std::vector<snappy_hip_ctx*> g_pool;
constexpr size_t kPoolKeep = 4;  // idle contexts kept (a large call uses three at once)

struct CtxLease {
  snappy_hip_ctx* c = nullptr;
  int status = SNAPPY_HIP_OK;
  CtxLease() {
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      if (!g_pool.empty()) {
        c = g_pool.back();
        g_pool.pop_back();
      }
    }
    if (!c) {
      int dev = 0;
      if (const char* e = getenv("SNAPPY_HIP_DEVICE")) dev = atoi(e);
      status = snappy_hip_ctx_create(&c, dev);
      if (status) c = nullptr;
    }
    if (c) guard = new DeviceGuard(c->device);
  }
  ~CtxLease() {
    delete guard;
    if (c) {
      // an idle context keeps its (grow-only) workspace: the pool keeps at most kPoolKeep of them, the rest
      // give their memory back (snappy_hip_release_pool() frees the idle ones at any time)
      bool keep;
      {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        keep = g_pool.size() < kPoolKeep;
        if (keep) g_pool.push_back(c);
      }
      if (!keep) snappy_hip_ctx_destroy(c);
    }
  }
  CtxLease(const CtxLease&) = delete;
  CtxLease& operator=(const CtxLease&) = delete;

 private:
  DeviceGuard* guard = nullptr;
};

// A caller's buffer, page-locked in place for the duration of a call: copies from and to it are
// then real DMA transfers that overlap with each other and with kernels (measured on the GPU box:
// 56 GB/s each way at once, against 28 GB/s each way from pageable memory; registering 512 MiB takes
// 5 ms).  If the range cannot be registered (it already is, for instance) copies fall back to the
// runtime's pageable path.
constexpr size_t kPinMin = 1u << 20;
// SNAPPY_HIP_PIN_HOST (read once): 0 = the runtime's pageable copies (default: measured fastest, INTEGRATION.md
// 6), 1 = page-lock the caller's whole buffers for the call, 2 = batch by batch, 3 = every bulk copy through the
// contexts' page-locked staging rings (stage_* below), 4 = only the smaller direction of a call through the ring.
inline int pin_mode() {
  static const int mode = [] {
    const char* e = getenv("SNAPPY_HIP_PIN_HOST");
    return e ? atoi(e) : 0;
  }();
  return mode;
}
struct HostPin {
  void* p = nullptr;
  HostPin(const void* ptr, size_t n, int when = 1) {
    if (pin_mode() == when && n >= kPinMin && hipHostRegister(const_cast<void*>(ptr), n, hipHostRegisterDefault) == hipSuccess)
      p = const_cast<void*>(ptr);
    else
      (void)hipGetLastError();
  }
  ~HostPin() {
    if (p) (void)hipHostUnregister(p);
  }
  HostPin(const HostPin&) = delete;
  HostPin& operator=(const HostPin&) = delete;
};

// ---- page-locked staging of the bulk copies (SNAPPY_HIP_PIN_HOST = 3 / 4; NOT the default) ----------------
// Every pooled context can own a ring of kStageSlots page-locked pieces, allocated once: an upload is "host
// threads copy a piece into the ring, the DMA engine takes it from there", piece after piece, the copy of piece
// k + 1 running beside the transfer of piece k; a download the same the other way round.  Measured (round 3,
// 1 GiB): the runtime's pageable path moves 56 GB/s one way on this box (its own staging), a
// host thread copies 14 GB/s, and the ring -- whole (3) or for the smaller direction only (4) -- is slower than
// or equal to the pageable calls for every entry point but compress() (+9 %): it stays an option, off by default.
constexpr size_t kStagePiece = 16u << 20;
constexpr int kStageSlots = 4;
constexpr size_t kStageMin = 1u << 20;  // smaller copies: the direct call (latency, not bandwidth, matters there)

class CopyPool {  // parallel memcpy; the threads live as long as the process (never joined: no static destructor)
 public:
  static CopyPool& get() {
    static CopyPool* p = new CopyPool();
    return *p;
  }
  void copy(void* dst, const void* src, size_t n) {
    constexpr size_t kGrain = 2u << 20;
    if (n < 2 * kGrain || n_threads_ == 0) {
      memcpy(dst, src, n);
      return;
    }
    size_t parts = n / kGrain;
    if (parts > (size_t)n_threads_ + 1) parts = (size_t)n_threads_ + 1;
    const size_t per = ((n + parts - 1) / parts + 63) & ~(size_t)63;
    Job job;
    job.left = (int)parts - 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t k = 1; k < parts; k++) {
        const size_t lo = k * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) {
          job.left--;
          continue;
        }
        q_.push_back({(uint8_t*)dst + lo, (const uint8_t*)src + lo, hi - lo, &job});
      }
    }
    cv_.notify_all();
    memcpy(dst, src, per < n ? per : n);
    std::unique_lock<std::mutex> lk(mu_);
    job.cv.wait(lk, [&] { return job.left <= 0; });
  }

 private:
  struct Job {
    int left = 0;
    std::condition_variable cv;
  };
  struct Task {
    uint8_t* d;
    const uint8_t* s;
    size_t n;
    Job* job;
  };
  CopyPool() {
    unsigned hw = std::thread::hardware_concurrency();
    n_threads_ = hw >= 16 ? 7 : (hw >= 4 ? (int)hw / 2 - 1 : 0);
    if (const char* e = getenv("SNAPPY_HIP_COPY_THREADS")) n_threads_ = atoi(e) > 0 ? atoi(e) - 1 : 0;
    for (int i = 0; i < n_threads_; i++) {
      try {
        std::thread([this] { run(); }).detach();
      } catch (...) {
        n_threads_ = i;
        break;
      }
    }
  }
  void run() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !q_.empty(); });
        t = q_.front();
        q_.pop_front();
      }
      memcpy(t.d, t.s, t.n);
      std::lock_guard<std::mutex> lk(mu_);
      if (--t.job->left <= 0) t.job->cv.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Task> q_;
  int n_threads_ = 0;
};

// SNAPPY_HIP_PIN_HOST = 3 (NOT the default, which is 0): the staging ring.  (0: the runtime's pageable copies; 1, 2: page-lock the
// caller's buffers, see HostPin.)
inline bool stage_ready(snappy_hip_ctx* c, bool small_side) {
  // 3: every bulk copy through the ring; 4: only the smaller direction of a call (the compressed side), the
  // larger one stays with the runtime's pageable path -- the two then run side by side, which two pageable
  // copies do not
  if (!(pin_mode() == 3 || (pin_mode() == 4 && small_side)) || c->stage_failed) return false;
  if (c->stage) return true;
  if (hipHostMalloc((void**)&c->stage, kStagePiece * kStageSlots, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    c->stage = nullptr;
    c->stage_failed = true;
    return false;
  }
  for (auto& e : c->stage_ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      c->stage_failed = true;
      return false;
    }
  return true;
}

inline int stage_wait(snappy_hip_ctx* c, int slot) {  // the ring piece is free again
  if (c->stage_busy[slot]) {
    HIP_TRY(hipEventSynchronize(c->stage_ev[slot]));
    c->stage_busy[slot] = false;
  }
  return SNAPPY_HIP_OK;
}

// d_dst[0 .. n) = src[0 .. n) (host, pageable), enqueued on s: returns when the last piece is in the ring (its
// transfer may still be under way; what is launched on s afterwards is ordered behind it).
int stage_upload(snappy_hip_ctx* c, void* d_dst, const uint8_t* src, size_t n, hipStream_t s, bool small_side) {
  if (n < kStageMin || !stage_ready(c, small_side)) {
    HIP_TRY(hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, s));
    return SNAPPY_HIP_OK;
  }
  int slot = 0;
  for (size_t off = 0; off < n; off += kStagePiece, slot = (slot + 1) % kStageSlots) {
    const size_t len = n - off < kStagePiece ? n - off : kStagePiece;
    int st = stage_wait(c, slot);
    if (st) return st;
    uint8_t* piece = c->stage + (size_t)slot * kStagePiece;
    CopyPool::get().copy(piece, src + off, len);
    HIP_TRY(hipMemcpyAsync((uint8_t*)d_dst + off, piece, len, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(c->stage_ev[slot], s));
    c->stage_busy[slot] = true;
  }
  return SNAPPY_HIP_OK;
}

// dst[0 .. n) (host, pageable) = d_src[0 .. n), behind what is enqueued on s; returns when the bytes are there.
int stage_download(snappy_hip_ctx* c, uint8_t* dst, const void* d_src, size_t n, hipStream_t s, bool small_side) {
  if (n < kStageMin || !stage_ready(c, small_side)) {
    HIP_TRY(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SNAPPY_HIP_OK;
  }
  const size_t pieces = (n + kStagePiece - 1) / kStagePiece;
  size_t issued = 0;
  for (size_t done = 0; done < pieces; done++) {
    // keep the ring full of transfers, then take the oldest piece out while the others are under way
    for (; issued < pieces && issued < done + kStageSlots; issued++) {
      const int slot = (int)(issued % kStageSlots);
      int st = stage_wait(c, slot);  // (a piece of an earlier upload)
      if (st) return st;
      const size_t off = issued * kStagePiece, len = n - off < kStagePiece ? n - off : kStagePiece;
      HIP_TRY(hipMemcpyAsync(c->stage + (size_t)slot * kStagePiece, (const uint8_t*)d_src + off, len, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipEventRecord(c->stage_ev[slot], s));
      c->stage_busy[slot] = true;
    }
    const int slot = (int)(done % kStageSlots);
    int st = stage_wait(c, slot);
    if (st) return st;
    const size_t off = done * kStagePiece, len = n - off < kStagePiece ? n - off : kStagePiece;
    CopyPool::get().copy(dst + off, c->stage + (size_t)slot * kStagePiece, len);
  }
  return SNAPPY_HIP_OK;
}

// Runs fn(batch index, context) for every batch: one batch inline, several on up to three worker
// threads, each with a context of its own -- the upload of one batch, the kernels of another and the
// download of a third then overlap.  Returns the first non-zero status.
template <typename F>
int run_batches(size_t n_batches, snappy_hip_ctx* own, F fn) {
  if (n_batches <= 1) return n_batches ? fn((size_t)0, own) : SNAPPY_HIP_OK;
  const size_t n_workers = n_batches < 3 ? n_batches : 3;
  std::atomic<size_t> next{0};
  std::atomic<int> first_err{SNAPPY_HIP_OK};
  std::vector<std::string> errs(n_workers);
  auto work = [&](size_t w, snappy_hip_ctx* c) {
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= n_batches) break;
      const int st = fn(b, c);
      if (st) {
        int exp = SNAPPY_HIP_OK;
        if (first_err.compare_exchange_strong(exp, st)) errs[w] = g_last_error;
      }
    }
  };
  std::vector<std::thread> threads;
  for (size_t w = 1; w < n_workers; w++) {
    try {
      threads.emplace_back([&, w] {
        CtxLease lease;
        if (!lease.c) {
          int exp = SNAPPY_HIP_OK;
          if (first_err.compare_exchange_strong(exp, lease.status)) errs[w] = g_last_error;
          return;
        }
        work(w, lease.c);
      });
    } catch (...) {  // (no thread to be had: the batches go to the workers there are; nothing crosses the C boundary)
      break;
    }
  }
  work(0, own);
  for (auto& t : threads) t.join();
  const int st = first_err.load();
  if (st)
    for (auto& e : errs)
      if (!e.empty()) g_last_error = e;
  return st;
}

// Encode `n` host bytes as units of 65 536 bytes and land the packed stream (from byte `base`) in
// out[base ..]; *total = end offset.  The input goes in batches of 64 MiB: a batch is uploaded,
// encoded and packed on one context while its neighbours are on theirs (run_batches); a batch's
// place in the output is the sum of the batches in front of it -- the reference's serial
// `written += ...` (snappy.nim:59-62, :149-153) -- so its download waits for their sizes only.
int encode_host(snappy_hip_ctx* own, const uint8_t* in, size_t n, int unit, uint8_t* out, size_t cap,
                uint64_t base, uint64_t* total) {
  const uint64_t nb = (n + kMaxBlockLen - 1) / kMaxBlockLen;
  const uint64_t kHostBatchBlocks = host_batch_blocks();
  const size_t n_batches = (size_t)((nb + kHostBatchBlocks - 1) / kHostBatchBlocks);
  const uint64_t bound = base + nb * (uint64_t)(kMaxCompressedBlockLen + 16);  // what can be produced at most
  HostPin pin_in(in, n), pin_out(out, (size_t)(bound < cap ? bound : cap));
  std::vector<uint64_t> sizes(n_batches, 0);
  std::vector<char> ready(n_batches, 0);
  std::mutex mu;
  std::condition_variable cv;
  bool failed = false;
  auto fn = [&](size_t b, snappy_hip_ctx* c) -> int {
    auto fail = [&](int st) {
      std::lock_guard<std::mutex> lk(mu);
      failed = true;
      cv.notify_all();
      return st;
    };
    const uint64_t b0 = b * kHostBatchBlocks;
    const uint64_t cnt = nb - b0 < kHostBatchBlocks ? nb - b0 : kHostBatchBlocks;
    const size_t in_lo = (size_t)(b0 * kMaxBlockLen);
    const size_t in_n = n - in_lo < cnt * kMaxBlockLen ? n - in_lo : (size_t)(cnt * kMaxBlockLen);
    void *d_in, *d_slots, *d_sizes, *d_offsets, *d_out;
    int st;
    if ((st = ws_get(c, 0, in_n + 64, &d_in))) return fail(st);
    if ((st = ws_get(c, 1, cnt * (size_t)kSlotStride, &d_slots))) return fail(st);
    if ((st = ws_get(c, 2, cnt * 4, &d_sizes))) return fail(st);
    if ((st = ws_get(c, 3, (cnt + 1) * 8, &d_offsets))) return fail(st);
    if ((st = ws_get(c, 4, cnt * (size_t)kSlotStride + 64, &d_out))) return fail(st);
    hipStream_t s = c->stream;
    HostPin pin_in(in + in_lo, in_n, 2);
    if ((st = stage_upload(c, d_in, in + in_lo, in_n, s, false))) return fail(st);
    if ((st = snappy_hip_encode_blocks_d(c, (const uint8_t*)d_in, in_n, kMaxBlockLen, unit, (uint8_t*)d_slots,
                                         kSlotStride, (uint32_t*)d_sizes, s)))
      return fail(st);
    if ((st = snappy_hip_pack_d(c, (const uint8_t*)d_slots, kSlotStride, (const uint32_t*)d_sizes, cnt, 0,
                                (uint8_t*)d_out, (uint64_t*)d_offsets, s)))
      return fail(st);
    uint64_t end = 0;
    if (hipMemcpyAsync(&end, (uint64_t*)d_offsets + cnt, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return fail(SNAPPY_HIP_DEVICE_ERROR);
    uint64_t at = base;
    {
      std::unique_lock<std::mutex> lk(mu);
      sizes[b] = end;
      ready[b] = 1;
      cv.notify_all();
      cv.wait(lk, [&] {
        if (failed) return true;
        for (size_t j = 0; j < b; j++)
          if (!ready[j]) return false;
        return true;
      });
      if (failed) return SNAPPY_HIP_DEVICE_ERROR;
      for (size_t j = 0; j < b; j++) at += sizes[j];
    }
    if (at + end > cap) {
      g_last_error = "internal: packed stream exceeds the caller's bound";
      return fail(SNAPPY_HIP_DEVICE_ERROR);
    }
    HostPin pin_out(out + at, (size_t)end, 2);
    if ((st = stage_download(c, out + at, d_out, (size_t)end, s, true))) return fail(st);
    return SNAPPY_HIP_OK;
  };
  const int st = run_batches(n_batches, own, fn);
  if (st) return st;
  uint64_t end = base;
  for (size_t j = 0; j < n_batches; j++) end += sizes[j];
  *total = end;
  return SNAPPY_HIP_OK;
}

struct HostUnit {
  uint64_t in_off;
  uint32_t in_len;
  uint64_t out_off;
  uint32_t out_cap;
  uint8_t kind;
};

// Run units on the device: input bytes in[0..n) are uploaded once, outputs land in the context's
// output buffer (out_bytes of it), the first copy_bytes of which are copied to out; per-unit
// status / length / crc come back in the vectors.
int decode_host(snappy_hip_ctx* c, const uint8_t* in, size_t n, const std::vector<HostUnit>& units,
                uint8_t* out, size_t out_bytes, bool want_crc, std::vector<uint32_t>* status,
                std::vector<uint32_t>* out_len, std::vector<uint32_t>* crc, size_t copy_bytes) {
  int st;
  const size_t nu = units.size();
  status->assign(nu, 0);
  out_len->assign(nu, 0);
  crc->assign(nu, 0);
  if (nu == 0) return SNAPPY_HIP_OK;
  // stored chunks (plain copies) are put behind the compressed units, so that those can take the
  // indexed decoder as one uniform batch; results are mapped back below
  std::vector<uint32_t> perm(nu);
  size_t n_front = 0;
  {
    size_t back = nu;
    for (size_t i = 0; i < nu; i++)
      if (units[i].kind != (uint8_t)kUnitStored) perm[n_front++] = (uint32_t)i;
    for (size_t i = nu; i-- > 0;)
      if (units[i].kind == (uint8_t)kUnitStored) perm[--back] = (uint32_t)i;
  }
  std::vector<uint64_t> io(nu), oo(nu);
  std::vector<uint32_t> il(nu), oc(nu);
  std::vector<uint8_t> kd(nu);
  for (size_t k = 0; k < nu; k++) {
    const HostUnit& un = units[perm[k]];
    io[k] = un.in_off;
    il[k] = un.in_len;
    oo[k] = un.out_off;
    oc[k] = un.out_cap;
    kd[k] = un.kind;
  }
  void *d_in, *d_out, *d_io, *d_il, *d_oo, *d_oc, *d_ol, *d_st, *d_kd, *d_crc;
  if ((st = ws_get(c, 0, n + 64, &d_in))) return st;
  if ((st = ws_get(c, 4, out_bytes + 64, &d_out))) return st;
  if ((st = ws_get(c, 3, nu * 8, &d_io))) return st;
  if ((st = ws_get(c, 2, nu * 4, &d_il))) return st;
  if ((st = ws_get(c, 5, nu * 8, &d_oo))) return st;
  if ((st = ws_get(c, 6, nu * 4, &d_oc))) return st;
  if ((st = ws_get(c, 7, nu * 4, &d_ol))) return st;
  if ((st = ws_get(c, 8, nu * 4, &d_st))) return st;
  if ((st = ws_get(c, 9, nu, &d_kd))) return st;
  if ((st = ws_get(c, 10, nu * 4, &d_crc))) return st;
  hipStream_t s = c->stream;
  if ((st = stage_upload(c, d_in, in, n, s, true))) return st;
  HIP_TRY(hipMemcpyAsync(d_io, io.data(), nu * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_il, il.data(), nu * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_oo, oo.data(), nu * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_oc, oc.data(), nu * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_kd, kd.data(), nu, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_ol, 0, nu * 4, s));
  // the compressed units take the indexed decoder when they are all of one kind; stored chunks
  // (and mixed kinds) take the one-pass kernel with per-unit kinds
  bool uniform = n_front > 0;
  for (size_t i = 1; i < n_front; i++) uniform = uniform && kd[i] == kd[0];
  const size_t n_a = uniform ? n_front : 0;  // units [0, n_a): uniform batch; [n_a, nu): per-unit kinds
  if (n_a && (st = decode_d(c, (const uint8_t*)d_in, (const uint64_t*)d_io, (const uint32_t*)d_il, n_a,
                            (int)kd[0], nullptr, (uint8_t*)d_out, (const uint64_t*)d_oo,
                            (const uint32_t*)d_oc, (uint32_t*)d_ol, (uint32_t*)d_st, true, s,
                            want_crc ? (uint32_t*)d_crc : nullptr)))
    return st;
  if (n_a < nu && (st = decode_d(c, (const uint8_t*)d_in, (const uint64_t*)d_io + n_a,
                                 (const uint32_t*)d_il + n_a, nu - n_a, 0, (const uint8_t*)d_kd + n_a,
                                 (uint8_t*)d_out, (const uint64_t*)d_oo + n_a, (const uint32_t*)d_oc + n_a,
                                 (uint32_t*)d_ol + n_a, (uint32_t*)d_st + n_a, true, s,
                                 want_crc ? (uint32_t*)d_crc + n_a : nullptr)))
    return st;
  // (want_crc: the CRC of what each unit produced -- decoded bytes / stored bytes, snappy.nim:231,
  // :245 -- came with the decode)
  if (want_crc) HIP_TRY(hipMemcpyAsync(crc->data(), d_crc, nu * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(status->data(), d_st, nu * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(out_len->data(), d_ol, nu * 4, hipMemcpyDeviceToHost, s));
  if (copy_bytes && (st = stage_download(c, out, d_out, copy_bytes, s, false))) return st;
  HIP_TRY(hipStreamSynchronize(s));
  {  // back to the caller's unit order
    std::vector<uint32_t> t(nu);
    for (std::vector<uint32_t>* v : {status, out_len, crc}) {
      for (size_t k = 0; k < nu; k++) t[perm[k]] = (*v)[k];
      *v = t;
    }
  }
  return SNAPPY_HIP_OK;
}

// Block starts of one raw buffer by the speculative parallel walk of split_kernels.h.  d_tags = the
// tag stream (behind the varint) in device memory.  Returns 0 (d_blk[k] = stream position of block
// k's first element for every k -- if the chain was complete: d_bad[0] bit 8 says when it was not), a status > 0
// (the walk saw the whole stream: its verdict stands), or -1: not applicable (the chain is not complete within the
// looks, an invalid or foreign element lies on it, a candidate list overflowed, an element straddles a block
// boundary).
// looks == false: everything is enqueued without a look from the host -- the bulk launch, four tail rounds (one with an
// empty queue is a load per wave), the marking, the placement; the verdict kernels note whether the root's chain
// reached the stream's end, and the caller, who looks once behind the decode, comes back with looks == true if it did
// not: the procedure from its start, the host reading every round's queue length (up to twelve rounds) before it marks.
int split_blocks_spec(snappy_hip_ctx* c, const uint8_t* d_tags, uint32_t n_tags, uint64_t len, size_t nblk,
                      uint32_t* d_blk, hipStream_t s, uint32_t* d_ol, uint32_t* d_bad, const uint64_t** d_total,
                      const uint32_t** d_flags, bool looks) {
  constexpr uint32_t kRoundsBlind = 4;
  const uint32_t nseg = (n_tags + kSplitSeg - 1) / kSplitSeg;
  const uint32_t nwg = (nseg + kSplitWg - 1) / kSplitWg;  // (the bulk launch: one wave and 16 KiB of staged stream each)
  const size_t nodes = (size_t)nseg * kSplitCand;
  void* base;
  // per node: ent, ext, ob, two jump tables (4 bytes each; the queues until the marking), reach (1); per segment: entry,
  // outb (4), out_at (8), the first walk's checkpoints (8), entry, exit, output bytes (4)
  int st = ws_get(c, 13, nodes * 21 + (size_t)nseg * 36 + (nodes / 256 + 1) * 4 + 8 + 64 + 64 + 8, &base);
  if (st) return st;
  uint8_t* q = (uint8_t*)base;
  uint64_t* out_at = (uint64_t*)q;
  q += ((size_t)nseg + 1) * 8;
  uint64_t* cp = (uint64_t*)q;
  q += (size_t)nseg * 8;
  uint32_t* ent = (uint32_t*)q;
  q += nodes * 4;
  uint32_t* ext = (uint32_t*)q;
  q += nodes * 4;
  uint32_t* ob = (uint32_t*)q;
  q += nodes * 4;
  uint32_t* jump[2];
  jump[0] = (uint32_t*)q, q += nodes * 4;
  jump[1] = (uint32_t*)q, q += nodes * 4;
  uint32_t* entry = (uint32_t*)q;
  q += (size_t)nseg * 4;
  uint32_t* outb = (uint32_t*)q;
  q += (size_t)nseg * 4;
  uint32_t* f_entry = (uint32_t*)q;
  q += (size_t)nseg * 4;
  uint32_t* f_code = (uint32_t*)q;
  q += (size_t)nseg * 4;
  uint32_t* f_ob = (uint32_t*)q;
  q += (size_t)nseg * 4;
  uint32_t* counters = (uint32_t*)q;  // [0] candidates added, [1] overflow, [2 + r] round r's queue length
  q += 64;
  uint32_t* flags = (uint32_t*)q;
  q += 32;
  uint32_t* blk_any = (uint32_t*)q;  // per 256 ids of the marking: a node among them (split_succ_kernel)
  q += (nodes / 256 + 1) * 4;
  uint8_t* reach = q;
  {  // ent and ext: no candidates but the root, nothing walked; counters, flags; block starts, lengths, verdict
    const uint32_t ig = (uint32_t)((nodes * 2 + 255) / 256 < 1024 ? (nodes * 2 + 255) / 256 : 1024);
    LAUNCH(split_init_kernel, dim3(ig), dim3(256), 0, s, ent, (uint64_t)nodes * 2, counters, d_blk, (uint32_t)(nblk + 1), d_ol,
           (uint32_t)nblk, d_bad);
  }
  SplitParams sp{};
  sp.in = d_tags;
  sp.n = n_tags;
  sp.nseg = nseg;
  sp.ent = ent;
  sp.ext = ext;
  sp.ob = ob;
  sp.counters = counters;
  sp.cp = cp;
  sp.f_entry = f_entry;
  sp.f_code = f_code;
  sp.f_ob = f_ob;
  sp.q_cap = (uint32_t)nodes;
  sp.entry = entry;
  sp.outb = outb;
  sp.flags = flags;
  sp.out_at = out_at;
  sp.blk_in = d_blk;
  sp.nblk = (uint32_t)nblk;
  sp.bad = d_bad;
  sp.budget = kSplitBudget;
  sp.hops = kSplitHops;
  if (const char* e = dbg_env("SNAPPY_HIP_SPLIT_KNOBS")) {  // DEBUG: "budget,hops"
    unsigned b = kSplitBudget, h = kSplitHops;
    sscanf(e, "%u,%u", &b, &h);
    sp.budget = b, sp.hops = h;
  }
  const uint32_t grid = (nseg + 255) / 256;
  const uint32_t ngrid = (uint32_t)((nodes + 255) / 256);
  HIP_TRY(hipFuncSetAttribute((const void*)split_bulk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplitStage));
  HIP_TRY(hipFuncSetAttribute((const void*)split_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplitStage));
  HIP_TRY(hipFuncSetAttribute((const void*)split_locate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplitStage));
  {
    LaunchTimer lt(c, s, 7);
    sp.q_out = jump[0];
    sp.q_out_count = counters + 2;
    LAUNCH(split_bulk_kernel, dim3(nwg), dim3(kSplitWg), kSplitStage, s, sp);
  }
  if (dbg_env("SNAPPY_HIP_STATS")) {  // DEBUG: the bulk launch's phases (debug builds time them)
    uint32_t h[8];
    HIP_TRY(hipMemcpyAsync(h, flags, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    fprintf(stderr, "SPLIT bulk, us per wave: stage %.1f first walk %.1f hand on + summary %.1f second walk %.1f push %.1f (%u waves)\n",
            h[3] / 100.0 / nwg, h[4] / 100.0 / nwg, h[5] / 100.0 / nwg, h[6] / 100.0 / nwg, h[7] / 100.0 / nwg, nwg);
    HIP_TRY(hipMemsetAsync(flags + 3, 0, 20, s));
  }
  const uint32_t tgrid = nwg < 2048 ? nwg : 2048;
  for (uint32_t r = 0; r + 1 < kSplitMaxRounds; r++) {
    if (!looks && r == kRoundsBlind) break;
    if (looks) {  // (the fallback's pace: a look a round)
      uint32_t h_n = 0;
      HIP_TRY(hipMemcpyAsync(&h_n, counters + 2 + r, 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (dbg_env("SNAPPY_HIP_STATS")) fprintf(stderr, "SPLIT round %u: %u nodes queued\n", r, h_n);  // DEBUG
      if (h_n == 0) break;
    }
    LaunchTimer lt(c, s, 7);
    sp.q_in = jump[r & 1];
    sp.q_in_count = counters + 2 + r;
    sp.q_out = jump[(r + 1) & 1];
    sp.q_out_count = counters + 3 + r;
    LAUNCH(split_tail_kernel, dim3(tgrid), dim3(kSplitWg), kSplitStage, s, sp);
  }
  // is the real chain complete?  mark what the root reaches; its last pointer tells
  int steps = 1;  // kSplitJump-fold pointer jumps that cover a chain of nseg nodes
  for (uint64_t reach_n = kSplitJump; reach_n < (uint64_t)nseg + 1; reach_n *= kSplitJump) steps++;
  LAUNCH(split_succ_kernel, dim3(ngrid), dim3(256), 0, s, sp, jump[0], reach, blk_any);
  int cur = 0;
  for (int k = 0; k < steps; k++, cur ^= 1)
    LAUNCH(split_double_kernel, dim3(ngrid), dim3(256), 0, s, (uint32_t)nodes, (const uint32_t*)jump[cur], jump[cur ^ 1], reach,
           (const uint32_t*)blk_any);
  HIP_TRY(hipGetLastError());
  if (looks) {
    uint32_t h_root = 0, h_cnt[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&h_root, jump[cur], 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h_cnt, counters, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (dbg_env("SNAPPY_HIP_STATS"))  // DEBUG
      fprintf(stderr, "SPLIT look: root -> %08x, candidates added %u, overflow %u\n", h_root, h_cnt[0], h_cnt[1]);
    // an invalid or foreign element on the chain, a list that overflowed and may have dropped the candidate the chain
    // needs, rounds that never ended: for the serial walk to judge
    if (h_root != kSplitEnd) return -1;
  }
  LAUNCH(split_select_kernel, dim3(grid), dim3(256), 0, s, sp, (const uint8_t*)reach, (const uint32_t*)jump[cur]);
  {  // out_at = exclusive prefix sum of outb (tile sums live in the jump tables, which are free now)
    const uint32_t tiles = (nseg + kSplitTile - 1) / kSplitTile;
    uint32_t* tile_sum = jump[cur ^ 1];
    uint64_t* tile_base = (uint64_t*)ext;  // [tiles + 1] (nodes * 4 bytes >= that; the walks' exits are not needed any more)
    LAUNCH(split_tile_sums_kernel, dim3(tiles), dim3(256), 0, s, (const uint32_t*)outb, nseg, tile_sum);
    LAUNCH(scan_sizes_kernel, dim3(1), dim3(kScanThreads), 0, s, (const uint32_t*)tile_sum, (uint64_t)tiles, (uint64_t)0, tile_base);
    LAUNCH(split_tile_scan_kernel, dim3(tiles), dim3(256), 0, s, (const uint32_t*)outb, nseg, (const uint64_t*)tile_base, out_at);
  }
  // (the total -- snappy.nim:107-108 -- and the last walk's flags are judged on the device, by split_table_kernel: the
  // host looks once, behind the decode; a total that is not the declared length writes no block start beyond the table)
  (void)len;
  if (nblk > 1) LAUNCH(split_locate_kernel, dim3((uint32_t)((nblk - 1 + kSplitWg - 1) / kSplitWg)), dim3(kSplitWg), kSplitStage, s, sp);
  HIP_TRY(hipGetLastError());
  *d_total = out_at + nseg;
  *d_flags = flags;
  return SNAPPY_HIP_OK;
}

// uncompress() of a raw buffer that decodes to more than one 64 KiB block.  The stream has no block
// delimiters (snappy.nim:49-62), so one wave first walks it and records where every 64 KiB of
// output starts (index_units_kernel<true>); the blocks are then decoded in parallel like
// independent units.  Returns -1 when that does not apply (an element or a copy crosses a 64 KiB
// output boundary, possible with foreign encoders): the caller then uses the serial kernel.
// (d_in_res / d_out_res: the buffer / the output are resident in device memory already)
int uncompress_split_host(snappy_hip_ctx* c, const uint8_t* in, size_t n, uint32_t hdr, uint64_t len,
                          uint8_t* out, size_t* written, const uint8_t* d_in_res = nullptr,
                          uint8_t* d_out_res = nullptr, hipStream_t on = nullptr) {
  if (len > (1ull << 31) || n >= (1ull << 31)) return -1;  // (positions carry a flag in bit 31: split_kernels.h)
  int st;
  const size_t nblk = (size_t)((len + kMaxBlockLen - 1) / kMaxBlockLen);
  void *d_in = const_cast<uint8_t*>(d_in_res), *d_out = d_out_res, *d_io, *d_il, *d_oo, *d_oc, *d_ol, *d_st, *d_blk,
       *d_one;
  if (!d_in_res && (st = ws_get(c, 0, n + 64, &d_in))) return st;
  if (!d_out_res && (st = ws_get(c, 4, len + 64, &d_out))) return st;
  if ((st = ws_get(c, 3, nblk * 8, &d_io))) return st;
  if ((st = ws_get(c, 2, nblk * 4, &d_il))) return st;
  if ((st = ws_get(c, 5, nblk * 8, &d_oo))) return st;
  if ((st = ws_get(c, 6, nblk * 4, &d_oc))) return st;
  if ((st = ws_get(c, 7, nblk * 4, &d_ol))) return st;
  if ((st = ws_get(c, 8, nblk * 4, &d_st))) return st;
  if ((st = ws_get(c, 9, (nblk + 1) * 4, &d_blk))) return st;
  if ((st = ws_get(c, 10, 64, &d_one))) return st;
  hipStream_t s = on ? on : c->stream;  // (the device-resident entry point passes the caller's stream on)
  if (!d_in_res && (st = stage_upload(c, d_in, in, n, s, true))) return st;
  uint32_t* d_bad = (uint32_t*)d_one + 6;  // d_one[6..8]: the split's verdict, the root's last pointer, overflow (split_kernels.h)
  // first the speculative parallel walk (split_kernels.h); the one-workgroup walk below is its fallback
  // (the marking is ~a dozen small launches, ~0.1 ms before anything is decoded: below ~200 KiB of stream the
  // one-workgroup walk, 1 GB/s, is there first)
  constexpr size_t kSpecMinStream = 192 << 10;
  const bool try_spec = !dbg_env("SNAPPY_HIP_NO_SPEC_SPLIT") && n - hdr >= kSpecMinStream;
  // attempt 0: the speculative walk with nothing looked at by the host before the decode is done; attempt 1, if its chain
  // was not complete: the same with looks; then (or at once, for a small stream) the one-workgroup walk
  for (int attempt = try_spec ? 0 : 2; attempt < 3; attempt++) {
    const uint64_t* d_total = nullptr;
    const uint32_t* d_flags = nullptr;
    const int spec = attempt == 2 ? -1
                                  : split_blocks_spec(c, (const uint8_t*)d_in + hdr, (uint32_t)(n - hdr), len, nblk, (uint32_t*)d_blk,
                                                      s, (uint32_t*)d_ol, d_bad, &d_total, &d_flags, attempt == 1);
    if (spec > 0) return spec;
    if (spec < 0 && attempt < 2) {
      attempt = 1;  // (given up: the one-workgroup walk starts from a clean table)
      continue;
    }
    if (spec < 0) {
      d_total = nullptr;
      d_flags = nullptr;
      HIP_TRY(hipMemsetAsync(d_blk, 0xff, (nblk + 1) * 4, s));
      HIP_TRY(hipMemsetAsync(d_bad, 0, 12, s));
      HIP_TRY(hipMemsetAsync(d_ol, 0, nblk * 4, s));
      // the splitter's one unit: [in_off u64 | in_len u32 | out_cap u32 | out_len u32 | status u32]
      struct {
        uint64_t in_off;
        uint32_t in_len, out_cap, out_len, status;
      } one = {0, (uint32_t)n, (uint32_t)len, 0, 0};
      HIP_TRY(hipMemcpyAsync(d_one, &one, sizeof(one), hipMemcpyHostToDevice, s));
      IndexParams ip{};
      ip.in = (const uint8_t*)d_in;
      ip.in_off = (const uint64_t*)d_one;
      ip.in_len = (const uint32_t*)((uint8_t*)d_one + 8);
      ip.out_cap = (const uint32_t*)((uint8_t*)d_one + 12);
      ip.out_len = (uint32_t*)((uint8_t*)d_one + 16);
      ip.status = (uint32_t*)((uint8_t*)d_one + 20);
      ip.n_units = 1;
      ip.unit = kUnitRaw;
      ip.blk_in = (uint32_t*)d_blk;
      unsigned long long* d_sdbg = nullptr;
      if (dbg_env("SNAPPY_HIP_STATS")) {  // DEBUG
        HIP_TRY(hipMalloc((void**)&d_sdbg, 64));
        HIP_TRY(hipMemsetAsync(d_sdbg, 0, 64, s));
        ip.idx = (uint32_t*)d_sdbg;
      }
      LAUNCH(index_units_kernel<true>, dim3(1), dim3(64 * kSplitWaves), 0, s, ip);
      if (d_sdbg) {
        unsigned long long h[8];
        HIP_TRY(hipMemcpyAsync(h, d_sdbg, 64, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const double nch = (double)((n + kChunk - 1) / kChunk) / kSplitWaves;
        fprintf(stderr, "SPLIT STATS wave 0, ticks per own chunk: tables %.0f, mail wait %.0f, chain %.0f\n", h[0] / nch,
                h[1] / nch, h[2] / nch);
        (void)hipFree(d_sdbg);
      }
      // the one-workgroup walk's verdict (the speculative walk has given its own)
      HIP_TRY(hipMemcpyAsync(&one, d_one, sizeof(one), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (one.status == kNeedsStreamKernel || one.status == kNeedsOnePass) return -1;
      if (one.status != kOk) return (int)one.status;  // the walk saw the whole stream: its verdict stands
    }
    // the blocks as units, on the device (no trip to the host in between); d_bad: a block without a start, ...
    LAUNCH(split_table_kernel, dim3((uint32_t)((nblk + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_blk, (uint32_t)nblk,
           (uint32_t)(n - hdr), hdr, len, (uint64_t*)d_io, (uint32_t*)d_il, (uint64_t*)d_oo, (uint32_t*)d_oc, d_bad, d_total,
           d_flags);
    if ((st = decode_d(c, (const uint8_t*)d_in, (const uint64_t*)d_io, (const uint32_t*)d_il, nblk,
                       (int)kUnitBody, nullptr, (uint8_t*)d_out, (const uint64_t*)d_oo,
                       (const uint32_t*)d_oc, (uint32_t*)d_ol, (uint32_t*)d_st, true, s)))
      return st;
    // every block to its full length?  (e.g. a copy that reaches into an earlier block: not)  One look at three words.
    LAUNCH(split_verdict_kernel, dim3((uint32_t)((nblk + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_st, (const uint32_t*)d_ol,
           (uint32_t)nblk, len, d_bad);
    uint32_t h_bad[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h_bad, d_bad, 12, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (attempt == 0 && (h_bad[0] & 8)) {  // the chain was not complete behind six launches (or never will be)
      if (dbg_env("SNAPPY_HIP_STATS"))  // DEBUG
        fprintf(stderr, "SPLIT without looks: root -> %08x, overflow %u: again, with looks\n", h_bad[1], h_bad[2]);
      if (h_bad[1] == kSplitBad) attempt = 1;  // (an invalid or foreign element on the chain: for the serial walk to judge)
      continue;
    }
    if (h_bad[0] & 2) return SNAPPY_HIP_INVALID_INPUT;  // (the speculative walk saw the whole stream: its verdict stands)
    if (h_bad[0]) return -1;
    break;
  }
  if (!d_out_res && (st = stage_download(c, out, d_out, (size_t)len, s, false))) return st;
  *written = (size_t)len;
  return SNAPPY_HIP_OK;
}

}  // namespace

// uncompress (snappy.nim:84-110) of ONE raw buffer resident in HBM into d_out (device).  A buffer of
// several blocks is split on the device (split_kernels.h) and its blocks are decoded in parallel;
// where that does not apply (a foreign encoder's elements straddle block boundaries) it is decoded
// by the serial whole-stream kernel.  *written (host): bytes produced.  Returns when done.
extern "C" int snappy_hip_uncompress_d(snappy_hip_ctx* c, const uint8_t* d_in, uint64_t n, uint8_t* d_out,
                                       uint64_t cap, uint64_t* written, void* stream) {
  *written = 0;
  DeviceGuard guard(c->device);
  hipStream_t s = pick_stream(c, stream);
  StreamTurn turn(c, s);
  if (n > 0xffffffffull) return SNAPPY_HIP_INVALID_INPUT;
  uint8_t hb[8] = {0};
  const size_t hn = n < 8 ? (size_t)n : 8;
  if (hn) HIP_TRY(hipMemcpyAsync(hb, d_in, hn, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  uint64_t len;
  const int hdr = varint_decode(hb, hn, 32, &len);  // snappy.nim:92-94
  if (hdr <= 0) return SNAPPY_HIP_INVALID_INPUT;
  if (cap < len) return SNAPPY_HIP_BUFFER_TOO_SMALL;  // snappy.nim:96-97
  if (len > kMaxBlockLen) {
    size_t w = 0;
    const int rs = uncompress_split_host(c, nullptr, (size_t)n, (uint32_t)hdr, len, nullptr, &w, d_in, d_out, s);
    if (rs >= 0) {
      *written = w;
      return rs;
    }
  }
  // one RAW unit: the block kernels, or the serial whole-stream kernel for more than 64 KiB
  void* d_u;
  int st = ws_get(c, 12, 64, &d_u);
  if (st) return st;
  struct {
    uint64_t in_off, out_off;
    uint32_t in_len, out_cap, out_len, status;
  } u = {0, 0, (uint32_t)n, (uint32_t)len, 0, 0};
  HIP_TRY(hipMemcpyAsync(d_u, &u, sizeof u, hipMemcpyHostToDevice, s));
  uint8_t* q = (uint8_t*)d_u;
  if ((st = decode_d(c, d_in, (const uint64_t*)q, (const uint32_t*)(q + 16), 1, kUnitRaw, nullptr, d_out,
                     (const uint64_t*)(q + 8), (const uint32_t*)(q + 20), (uint32_t*)(q + 24), (uint32_t*)(q + 28), true, s)))
    return st;
  HIP_TRY(hipMemcpyAsync(&u, d_u, sizeof u, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (u.status != kOk) return (int)u.status;
  *written = u.out_len;
  return SNAPPY_HIP_OK;
}

extern "C" void snappy_hip_release_pool(void) {
  std::vector<snappy_hip_ctx*> idle;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    idle.swap(g_pool);
  }
  for (snappy_hip_ctx* c : idle) snappy_hip_ctx_destroy(c);
}

extern "C" int snappy_hip_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                   size_t* written) {
  *written = 0;
  if ((uint64_t)n > 0xffffffffull) return SNAPPY_HIP_INVALID_INPUT;  // snappy.nim:41-42
  if ((uint64_t)cap < snappy_hip_max_compressed_len((uint32_t)n))    // snappy.nim:44-45
    return SNAPPY_HIP_BUFFER_TOO_SMALL;
  CtxLease lease;  // (an empty input still refuses to run without a device)
  if (!lease.c) return lease.status;
  const int hl = varint_encode_u32((uint32_t)n, out);  // snappy.nim:49-50
  uint64_t total = (uint64_t)hl;
  if (n) {
    int st = encode_host(lease.c, in, n, kUnitBody, out, cap, (uint64_t)hl, &total);
    if (st) return st;
  }
  *written = (size_t)total;
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_encode_block(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                       size_t* written) {
  *written = 0;
  if (n == 0 || n > kMaxBlockLen) return SNAPPY_HIP_INVALID_INPUT;  // encoder.nim:208-209
  if ((uint64_t)cap + 16 <= snappy_hip_max_compressed_len((uint32_t)n))  // encoder.nim:217
    return SNAPPY_HIP_BUFFER_TOO_SMALL;
  CtxLease lease;
  if (!lease.c) return lease.status;
  uint64_t total = 0;
  int st = encode_host(lease.c, in, n, kUnitBody, out, cap, 0, &total);
  if (st) return st;
  *written = (size_t)total;
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_encode_frame(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                       size_t* written) {
  *written = 0;
  if (n == 0 || n > kMaxBlockLen) return SNAPPY_HIP_INVALID_INPUT;  // encoder.nim:388
  if ((uint64_t)cap < snappy_hip_max_compressed_len((uint32_t)n))   // encoder.nim:392
    return SNAPPY_HIP_BUFFER_TOO_SMALL;
  CtxLease lease;
  if (!lease.c) return lease.status;
  uint64_t total = 0;
  int st = encode_host(lease.c, in, n, kUnitFrame, out, cap, 0, &total);
  if (st) return st;
  *written = (size_t)total;
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_compress_framed(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                          size_t* written) {
  *written = 0;
  if ((uint64_t)cap < snappy_hip_max_compressed_len_framed((int64_t)n))  // snappy.nim:139-140
    return SNAPPY_HIP_BUFFER_TOO_SMALL;
  CtxLease lease;
  if (!lease.c) return lease.status;
  memcpy(out, kFramingHeader, sizeof kFramingHeader);  // snappy.nim:142
  uint64_t total = sizeof kFramingHeader;
  if (n) {
    int st = encode_host(lease.c, in, n, kUnitFrame, out, cap, sizeof kFramingHeader, &total);
    if (st) return st;
  }
  *written = (size_t)total;
  return SNAPPY_HIP_OK;
}

extern "C" uint32_t snappy_hip_masked_crc32c(const uint8_t* buf, size_t n, int* status) {
  auto fail = [&](int st) -> uint32_t {
    if (status) *status = st;
    return 0;
  };
  if (n > 0xffffffffull) return fail(SNAPPY_HIP_INVALID_INPUT);  // crc32c.c:761 truncates; we refuse
  CtxLease lease;
  if (!lease.c) return fail(lease.status);
  snappy_hip_ctx* c = lease.c;
  int st;
  HostPin pin(buf, n);
  void *d_in, *d_off, *d_len, *d_crc;
  if ((st = ws_get(c, 0, n + 64, &d_in))) return fail(st);
  if ((st = ws_get(c, 3, 8, &d_off))) return fail(st);
  if ((st = ws_get(c, 2, 4, &d_len))) return fail(st);
  if ((st = ws_get(c, 10, 4, &d_crc))) return fail(st);
  uint64_t off = 0;
  uint32_t len = (uint32_t)n, crc = 0;
  hipStream_t s = c->stream;
  if (hipMemcpyAsync(d_in, buf, n, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_off, &off, 8, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(d_len, &len, 4, hipMemcpyHostToDevice, s) != hipSuccess)
    return fail(SNAPPY_HIP_DEVICE_ERROR);
  if ((st = snappy_hip_crc32c_d(c, (const uint8_t*)d_in, (const uint64_t*)d_off,
                                (const uint32_t*)d_len, 1, (uint32_t*)d_crc, s)))
    return fail(st);
  if (hipMemcpyAsync(&crc, d_crc, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return fail(SNAPPY_HIP_DEVICE_ERROR);
  if (status) *status = SNAPPY_HIP_OK;
  return crc;
}

extern "C" int snappy_hip_uncompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                     size_t* written) {
  *written = 0;
  uint64_t len;
  int hdr = varint_decode(in, n, 32, &len);  // snappy.nim:92-94
  if (hdr <= 0) return SNAPPY_HIP_INVALID_INPUT;
  if ((uint64_t)cap < len) return SNAPPY_HIP_BUFFER_TOO_SMALL;  // snappy.nim:96-97
  if (n > 0xffffffffull) return SNAPPY_HIP_INVALID_INPUT;       // unit lengths are 32-bit
  CtxLease lease;
  if (!lease.c) return lease.status;
  HostPin pin_in(in, n), pin_out(out, (size_t)len);
  if (len > kMaxBlockLen && !dbg_env("SNAPPY_HIP_NO_SPLIT")) {  // several blocks: split, then decode in parallel
    const int rs = uncompress_split_host(lease.c, in, n, (uint32_t)hdr, len, out, written);
    if (rs >= 0) return rs;
  }
  // the kernel re-parses the header: one RAW unit, output window = the declared length
  std::vector<HostUnit> units{{0, (uint32_t)n, 0, (uint32_t)len, (uint8_t)kUnitRaw}};
  std::vector<uint32_t> st, ol, crc;
  int rc = decode_host(lease.c, in, n, units, out, (size_t)len, false, &st, &ol, &crc, (size_t)len);
  if (rc) return rc;
  if (st[0] != kOk) return (int)st[0];
  *written = ol[0];
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_decode_all_tags(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                          size_t* written) {
  *written = 0;
  if (n == 0) return SNAPPY_HIP_OK;                    // decoder.nim:26-27
  if (cap == 0) return SNAPPY_HIP_BUFFER_TOO_SMALL;    // decoder.nim:29-30
  if (n > 0xffffffffull) return SNAPPY_HIP_INVALID_INPUT;
  const uint32_t cap32 = cap > 0xffffffffull ? 0xffffffffu : (uint32_t)cap;
  CtxLease lease;
  if (!lease.c) return lease.status;
  snappy_hip_ctx* c = lease.c;
  int rc;
  // The output length is not known up front.  A stream of n bytes cannot expand to more than
  // 64 bytes per 3 (copy2), so the device buffer is bounded by that; the unit's limit stays
  // the caller's capacity.  Only the bytes actually produced are copied back.
  const uint64_t bound = (uint64_t)n * 32 + 64;
  const size_t dev_out = (size_t)(bound < cap32 ? bound : cap32);
  std::vector<HostUnit> units{{0, (uint32_t)n, 0, cap32, (uint8_t)kUnitBody}};
  std::vector<uint32_t> st, ol, crc;
  rc = decode_host(c, in, n, units, out, dev_out, false, &st, &ol, &crc, 0);
  if (rc) return rc;
  if (st[0] != kOk) return (int)st[0];
  if (ol[0]) HIP_TRY(hipMemcpy(out, c->ws[4].p, ol[0], hipMemcpyDeviceToHost));
  *written = ol[0];
  return SNAPPY_HIP_OK;
}

extern "C" int snappy_hip_uncompress_framed(const uint8_t* in, size_t n, uint8_t* out, size_t cap,
                                            int check_header, int check_integrity,
                                            size_t* read_out, size_t* written_out) {
  *read_out = 0;
  *written_out = 0;
  if (n > 0xffffffffull * 16) return SNAPPY_HIP_INVALID_INPUT;
  size_t rd = 0, wr = 0;
  if (check_header) {  // snappy.nim:187-196
    if (n < sizeof kFramingHeader) return SNAPPY_HIP_INVALID_INPUT;
    if (memcmp(in, kFramingHeader, sizeof kFramingHeader) != 0) return SNAPPY_HIP_INVALID_INPUT;
    rd = sizeof kFramingHeader;
  }
  // ---- sequential header walk (host, like codec.nim:178-214): builds the unit list ---------
  struct Chunk {
    size_t hdr_at;     // offset of the chunk header
    size_t end_at;     // offset after the chunk
    uint32_t crc;      // stored masked CRC
    bool crc_only;     // stored chunk that is only checksummed (it ends the walk)
    int after;         // outcome once this chunk's CRC verified: -1 = continue
  };
  std::vector<HostUnit> units;
  std::vector<Chunk> chunks;
  int terminal = -1;          // status that ends the walk (-1: ran to the end of input)
  bool stop_ok = false;       // walk ended because the output is full: ok((read-4, written))
  size_t stop_rd = 0, stop_wr = 0;
  while (rd < n) {         // snappy.nim:199
    size_t remaining = n - rd;
    if (remaining < 4) {
      terminal = SNAPPY_HIP_INVALID_INPUT;
      break;
    }
    uint32_t hdr = (uint32_t)in[rd] | ((uint32_t)in[rd + 1] << 8) | ((uint32_t)in[rd + 2] << 16) |
                   ((uint32_t)in[rd + 3] << 24);
    uint8_t id = (uint8_t)hdr;
    size_t data_len = hdr >> 8;
    size_t hdr_at = rd;
    rd += 4;
    if (remaining - 4 < data_len) {  // snappy.nim:206-207
      terminal = SNAPPY_HIP_INVALID_INPUT;
      break;
    }
    if (id == 0x00) {  // snappy.nim:209-235
      if (data_len < 4) {
        terminal = SNAPPY_HIP_INVALID_INPUT;
        break;
      }
      uint32_t crc = (uint32_t)in[rd] | ((uint32_t)in[rd + 1] << 8) | ((uint32_t)in[rd + 2] << 16) |
                     ((uint32_t)in[rd + 3] << 24);
      size_t room = cap - wr;
      size_t max_out = room < kMaxBlockLen ? room : kMaxBlockLen;  // snappy.nim:215
      uint64_t ulen;
      int h = varint_decode(in + rd + 4, data_len - 4, 32, &ulen);  // uncompress, snappy.nim:92
      if (h <= 0) {
        terminal = SNAPPY_HIP_INVALID_INPUT;
        break;
      }
      if ((uint64_t)max_out < ulen) {  // bufferTooSmall inside the chunk, snappy.nim:219-227
        uint64_t u64;
        if (varint_decode(in + rd + 4, data_len - 4, 64, &u64) <= 0 || u64 > kMaxBlockLen) {
          terminal = SNAPPY_HIP_INVALID_INPUT;
        } else {
          stop_ok = true;
          stop_rd = hdr_at;
          stop_wr = wr;
        }
        break;
      }
      units.push_back({rd + 4, (uint32_t)(data_len - 4), wr, (uint32_t)ulen, (uint8_t)kUnitRaw});
      chunks.push_back({hdr_at, rd + data_len, crc, false, -1});
      wr += (size_t)ulen;
    } else if (id == 0x01) {  // snappy.nim:237-257
      if (data_len < 4) {
        terminal = SNAPPY_HIP_INVALID_INPUT;
        break;
      }
      uint32_t crc = (uint32_t)in[rd] | ((uint32_t)in[rd + 1] << 8) | ((uint32_t)in[rd + 2] << 16) |
                     ((uint32_t)in[rd + 3] << 24);
      size_t ul = data_len - 4;
      // the reference verifies the CRC BEFORE the size checks (snappy.nim:244-254)
      if (ul > kMaxBlockLen || ul > cap - wr) {
        const int after = ul > kMaxBlockLen ? SNAPPY_HIP_INVALID_INPUT : -2;  // -2: output full
        if (check_integrity) {  // checksum it on the device without delivering it
          units.push_back({rd + 4, (uint32_t)ul, 0, (uint32_t)ul, (uint8_t)kUnitStored});
          chunks.push_back({hdr_at, rd + data_len, crc, true, after});
        } else if (after == -2) {
          stop_ok = true;
          stop_rd = hdr_at;
          stop_wr = wr;
        } else {
          terminal = after;
        }
        break;
      }
      units.push_back({rd + 4, (uint32_t)ul, wr, (uint32_t)ul, (uint8_t)kUnitStored});
      chunks.push_back({hdr_at, rd + data_len, crc, false, -1});
      wr += ul;
    } else if (id < 0x80) {  // snappy.nim:259-260
      terminal = SNAPPY_HIP_UNKNOWN_CHUNK;
      break;
    }
    // 0x80..0xff skipped without validation, snappy.nim:262-263
    rd += data_len;
  }
  const size_t walk_rd = rd;

  // ---- device pass ---------------------------------------------------------------------------
  // In batches of about 64 MiB of output: a batch's slice of the stream is uploaded, its chunks are
  // decoded and checksummed, and its bytes are downloaded straight into the caller's buffer, on one
  // context, while the neighbouring batches are on theirs (run_batches) -- upload, kernels and
  // download overlap.  Chunks behind a failing one may thus have been delivered as well ("on error
  // output may have been partially written", snappy.nim:185); the verdict below is the reference's.
  const size_t deliver = wr;
  const size_t nu_all = units.size();
  for (size_t i = 0; i < nu_all; i++)
    if (chunks[i].crc_only) units[i].out_off = deliver;  // checksummed in scratch behind the batch's output
  CtxLease lease;
  if (!lease.c) return lease.status;
  HostPin pin_in(in, n), pin_out(out, deliver);
  std::vector<size_t> batch_lo;  // first unit of each batch
  {
    size_t acc_out = 0, acc_in = 0;
    for (size_t i = 0; i < nu_all; i++) {
      if (i == 0 || acc_out >= (64u << 20) || acc_in >= (64u << 20)) {
        batch_lo.push_back(i);
        acc_out = acc_in = 0;
      }
      acc_out += units[i].out_cap;
      acc_in += units[i].in_len;
    }
  }
  std::vector<uint32_t> st(nu_all), ol(nu_all), crc(nu_all);
  auto fn = [&](size_t b, snappy_hip_ctx* c) -> int {
    const size_t lo = batch_lo[b], hi = b + 1 < batch_lo.size() ? batch_lo[b + 1] : nu_all;
    const uint64_t in_lo = units[lo].in_off, in_hi = units[hi - 1].in_off + units[hi - 1].in_len;
    const uint64_t out_lo = units[lo].out_off;
    std::vector<HostUnit> part(units.begin() + lo, units.begin() + hi);
    uint64_t out_hi = out_lo;  // deliverable bytes end
    size_t scratch = 0;
    for (size_t k = 0; k < part.size(); k++) {
      part[k].in_off -= in_lo;
      if (chunks[lo + k].crc_only) {
        scratch = part[k].out_cap;
      } else {
        out_hi = part[k].out_off + part[k].out_cap;
      }
      part[k].out_off -= out_lo;
    }
    std::vector<uint32_t> s1, o1, c1;
    HostPin pin_in(in + in_lo, (size_t)(in_hi - in_lo), 2), pin_out(out + out_lo, (size_t)(out_hi - out_lo), 2);
    const int rc = decode_host(c, in + in_lo, (size_t)(in_hi - in_lo), part, out + out_lo,
                               (size_t)(out_hi - out_lo) + scratch, check_integrity != 0, &s1, &o1, &c1,
                               (size_t)(out_hi - out_lo));
    if (rc) return rc;
    for (size_t k = 0; k < part.size(); k++) {
      st[lo + k] = s1[k];
      ol[lo + k] = o1[k];
      crc[lo + k] = c1[k];
    }
    return SNAPPY_HIP_OK;
  };
  int rc = run_batches(batch_lo.size(), lease.c, fn);
  if (rc) return rc;
  auto deliver_prefix = [&](size_t) -> int { return SNAPPY_HIP_OK; };  // (delivered by the batches)

  // ---- first failure in stream order wins ------------------------------------------------------
  size_t ok_wr = 0;
  for (size_t i = 0; i < units.size(); i++) {
    const Chunk& ch = chunks[i];
    int result = -1;
    if (units[i].kind == kUnitRaw && st[i] != kOk) {
      result = SNAPPY_HIP_INVALID_INPUT;  // snappy.nim:228
    } else if (check_integrity && crc[i] != ch.crc) {
      result = SNAPPY_HIP_CRC_MISMATCH;  // snappy.nim:231-233, :244-246
    } else if (ch.crc_only) {
      result = ch.after;
    }
    if (result == -1) {
      ok_wr += ol[i];
      continue;
    }
    if ((rc = deliver_prefix(ok_wr))) return rc;  // chunks before the failing one were delivered
    if (result == -2) {                   // output full at this stored chunk, snappy.nim:253-254
      *read_out = ch.hdr_at;
      *written_out = ok_wr;
      return SNAPPY_HIP_OK;
    }
    return result;
  }
  if ((rc = deliver_prefix(deliver))) return rc;
  if (terminal >= 0) return terminal;
  if (stop_ok) {
    *read_out = stop_rd;
    *written_out = stop_wr;
    return SNAPPY_HIP_OK;
  }
  *read_out = walk_rd;
  *written_out = deliver;
  return SNAPPY_HIP_OK;
}
