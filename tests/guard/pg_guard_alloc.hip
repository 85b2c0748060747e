// Guard-page device allocator for the GPU parity tests (TEST INFRASTRUCTURE, never loaded by the package).
//
// Plugged into torch with torch.cuda.memory.CUDAPluggableAllocator (tests/guard/__init__.py, PG_GUARD=1): every
// tensor of a test run then lives in its OWN virtual-memory mapping,
//
//     [ unmapped guard | mapped pages ............................ | unmapped guard ]
//                        ^ canary bytes      ^ the tensor (flush against the far edge)
//
// so that an out-of-bounds access of a kernel is no longer absorbed by the caching allocator's 2-20 MB segments:
//   * a read or write past the flush edge hits an unmapped page -> "Memory access fault by GPU" at the launch that
//     did it (run with AMD_SERIALIZE_KERNEL=3 so the Python stack printed at the abort is the guilty call);
//   * a write on the slack side lands in the canary bytes, which are verified when the tensor is freed and by
//     pg_guard_check_all(); violations are counted and described (pg_guard_violations / pg_guard_report).
// PG_GUARD_SIDE=end (default) puts the tensor's END on the guard (overruns), =start its START (underruns).
// PG_GUARD_ALIGN (default 512) is the rounding of sizes / addresses: 512 is what torch's caching allocator
// guarantees in production; 16 is the strict setting (every byte past numel() is a fault).
//
// hipGraph capture: nothing may be mapped, synchronised or unmapped while a stream captures, so allocations made
// during a capture come from a plain arena (no guards) that is never recycled, and frees are deferred for ever —
// the eager warm-up iterations in front of every capture run the same kernels on guarded tensors.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

// Fills and checks go through KERNELS, not hipMemset / hipMemcpy: the copy engines' handling of pointers into
// hipMemMap'ed ranges at non-zero offsets proved unreliable on this runtime (the first version of this harness
// read back canaries that no kernel had touched as damaged, and missed a deliberate stray write).
__global__ void guard_fill_kernel(unsigned char* p, unsigned char v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// res[0] = damaged bytes, res[1] = first damaged index, res[2] = last damaged index, res[3..10] = first 8 damaged bytes' values (at first..)
__global__ void guard_check_kernel(const unsigned char* p, unsigned char v, size_t n, unsigned long long* res) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != v) {
      atomicAdd(&res[0], 1ull);
      atomicMin(&res[1], (unsigned long long)i);
      atomicMax(&res[2], (unsigned long long)i);
    }
}
__global__ void guard_peek_kernel(const unsigned char* p, size_t first, size_t last, unsigned long long* res) {
  const int t = threadIdx.x;
  if (t < 8 && first + t <= last) res[3 + t] = p[first + t];
  if (t >= 8 && t < 16 && last >= (size_t)(15 - t)) res[3 + t] = p[last - (15 - t)];
}
unsigned long long* g_res = nullptr;  // 19 words of plain device memory

struct Block {
  void* va;          // reserved range: END side [mapped | guard], START side [guard][mapped] (two reservations)
  size_t va_bytes;
  void* lead_guard;  // START side: the separately reserved guard granule in front of the mapping (or null)
  void* mapped;      // first mapped byte
  size_t mapped_bytes;
  hipMemGenericAllocationHandle_t handle;
  size_t user_bytes;   // rounded size handed to torch
  size_t asked_bytes;  // size torch asked for
  unsigned long serial;
};

std::mutex g_mu;
std::unordered_map<void*, Block> g_live;
size_t g_gran = 0;
size_t g_align = 512;
bool g_end_side = true;
bool g_poison = true;
int g_violations = 0;
long g_unguarded = 0;
long g_settle_retries = 0;  // fills of FRESH physical memory that did not stick (see settle())
std::unordered_multimap<size_t, hipMemGenericAllocationHandle_t> g_pool;  // released physical allocations by size  // START side: blocks whose leading guard reservation could not be placed
unsigned long g_serial = 0;
std::string g_report;
char* g_arena = nullptr;
size_t g_arena_bytes = 0, g_arena_used = 0;
const size_t kCanaryCheck = 64 << 10;  // canary bytes verified next to the tensor (the rest of the slack is filled too)
const unsigned char kCanary = 0xCB;

void die(const char* what, hipError_t e) {
  fprintf(stderr, "[pg_guard] %s failed: %s\n", what, hipGetErrorString(e));
  abort();
}
#define GCHECK(call)                        \
  do {                                      \
    hipError_t e_ = (call);                 \
    if (e_ != hipSuccess) die(#call, e_);   \
  } while (0)

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void init_once(int device) {
  if (g_gran) return;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  GCHECK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
  if (const char* a = getenv("PG_GUARD_ALIGN")) g_align = (size_t)atol(a);
  if (g_align < 16) g_align = 16;
  if (const char* s = getenv("PG_GUARD_SIDE")) g_end_side = strcmp(s, "start") != 0;
  if (const char* s = getenv("PG_GUARD_POISON")) g_poison = atoi(s) != 0;
  size_t arena_mb = 4096;
  if (const char* s = getenv("PG_GUARD_ARENA_MB")) arena_mb = (size_t)atol(s);
  g_arena_bytes = arena_mb << 20;
  GCHECK(hipMalloc((void**)&g_arena, g_arena_bytes));
  GCHECK(hipMalloc((void**)&g_res, 19 * sizeof(unsigned long long)));
  fprintf(stderr, "[pg_guard] active: granularity %zu B, align %zu, tensors flush against the %s guard, capture arena %zu MB\n",
          g_gran, g_align, g_end_side ? "END" : "START", arena_mb);
}

bool capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}

// Verifies the canary bytes of one block (device must be idle). Returns the number of damaged bytes.
size_t check_block(const Block& b, void* user) {
  size_t slack = b.mapped_bytes - b.user_bytes;
  size_t n = slack < kCanaryCheck ? slack : kCanaryCheck;
  if (!n) return 0;
  const unsigned char* from = g_end_side ? (unsigned char*)user - n : (unsigned char*)user + b.user_bytes;
  unsigned long long host[19] = {0, ~0ull, 0};
  GCHECK(hipMemcpy(g_res, host, sizeof host, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(guard_check_kernel, dim3(n >= 4096 ? 16 : 1), dim3(256), 0, nullptr, from, kCanary, n, g_res);
  GCHECK(hipMemcpy(host, g_res, 3 * sizeof host[0], hipMemcpyDeviceToHost));
  const size_t bad = (size_t)host[0];
  if (bad) {
    const size_t first = (size_t)host[1], last = (size_t)host[2];
    hipLaunchKernelGGL(guard_peek_kernel, dim3(1), dim3(64), 0, nullptr, from, first, last, g_res);
    GCHECK(hipMemcpy(host, g_res, sizeof host, hipMemcpyDeviceToHost));
    char line[384];
    long off0 = g_end_side ? (long)first - (long)n : (long)(b.user_bytes + first);
    long off1 = g_end_side ? (long)last - (long)n : (long)(b.user_bytes + last);
    int len = snprintf(line, sizeof line,
                       "allocation #%lu (%zu bytes asked, %zu given): %zu canary bytes overwritten, offsets %ld..%ld "
                       "relative to the tensor start; first bytes:",
                       b.serial, b.asked_bytes, b.user_bytes, bad, off0, off1);
    for (int i = 0; i < 8 && first + i <= last; i++) len += snprintf(line + len, sizeof line - len, " %02llx", host[3 + i]);
    len += snprintf(line + len, sizeof line - len, " last bytes:");
    for (int i = 8; i < 16; i++)
      if (last >= (size_t)(15 - i)) len += snprintf(line + len, sizeof line - len, " %02llx", host[3 + i]);
    snprintf(line + len, sizeof line - len, "\n");
    g_report += line;
    fprintf(stderr, "[pg_guard] VIOLATION %s", line);
    g_violations++;
  }
  return bad;
}

void fill(void* p, unsigned char v, size_t n);

// Fresh physical memory is cleared by the driver, and that clear can land AFTER the first kernels that touch the new
// mapping (observed: canaries written right after hipMemCreate + hipMemMap read back as zeros a moment later). A new
// allocation is therefore written and read back until two consecutive rounds, 200 us apart, find the pattern intact;
// physical allocations are then recycled through g_pool (no further clears, and no hipMemCreate per tensor).
void settle(void* p, size_t n) {
  int good = 0;
  for (int round = 0; round < 200 && good < 2; ++round) {
    const unsigned char pat = (unsigned char)(0xA0 + (round & 15));
    fill(p, pat, n);
    GCHECK(hipStreamSynchronize(nullptr));
    usleep(200);
    unsigned long long host[3] = {0, ~0ull, 0};
    GCHECK(hipMemcpy(g_res, host, sizeof host, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(guard_check_kernel, dim3(n >= 65536 ? 64 : 4), dim3(256), 0, nullptr, (const unsigned char*)p, pat, n, g_res);
    GCHECK(hipMemcpy(host, g_res, sizeof host, hipMemcpyDeviceToHost));
    if (host[0] == 0) ++good; else { good = 0; ++g_settle_retries; }
  }
  if (good < 2) fprintf(stderr, "[pg_guard] WARNING: a fresh %zu-byte allocation never kept its fill pattern\n", n);
}

void fill(void* p, unsigned char v, size_t n) {
  if (!n) return;
  size_t blocks = (n + 256 * 64 - 1) / (256 * 64);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(guard_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, nullptr, (unsigned char*)p, v, n);
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) void* pg_guard_malloc(ssize_t size, int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  GCHECK(hipSetDevice(device));
  init_once(device);
  if (size <= 0) size = 1;
  if (capturing(stream)) {
    size_t need = round_up((size_t)size, 512);
    if (g_arena_used + need > g_arena_bytes) {
      fprintf(stderr, "[pg_guard] capture arena exhausted (PG_GUARD_ARENA_MB)\n");
      abort();
    }
    void* p = g_arena + g_arena_used;
    g_arena_used += need;
    return p;
  }
  Block b = {};
  b.asked_bytes = (size_t)size;
  b.user_bytes = round_up((size_t)size, g_align);
  b.mapped_bytes = round_up(b.user_bytes, g_gran);
  b.serial = ++g_serial;
  // The mapping always starts at the BASE of its reservation (mapping at an offset inside a reservation is what the
  // copy engines mishandled). END side: one reservation [mapped | guard granule]. START side: the guard granule in
  // front is a reservation of its own, placed by reserving guard + mapped in one piece, releasing it and re-reserving
  // the two parts at the same addresses.
  if (g_end_side) {
    b.va_bytes = b.mapped_bytes + g_gran;
    GCHECK(hipMemAddressReserve(&b.va, b.va_bytes, g_gran, nullptr, 0));
    b.lead_guard = nullptr;
  } else {
    void* whole = nullptr;
    GCHECK(hipMemAddressReserve(&whole, b.mapped_bytes + g_gran, g_gran, nullptr, 0));
    GCHECK(hipMemAddressFree(whole, b.mapped_bytes + g_gran));
    void* guard = nullptr;
    if (hipMemAddressReserve(&guard, g_gran, g_gran, whole, 0) != hipSuccess || guard != whole) {
      (void)hipGetLastError();
      if (guard) GCHECK(hipMemAddressFree(guard, g_gran));
      guard = nullptr;
    }
    b.va_bytes = b.mapped_bytes;
    void* want = guard ? (char*)guard + g_gran : nullptr;
    GCHECK(hipMemAddressReserve(&b.va, b.va_bytes, g_gran, want, 0));
    if (guard && b.va != want) {  // no guard page in front of this block: canaries only
      GCHECK(hipMemAddressFree(guard, g_gran));
      guard = nullptr;
      g_unguarded++;
    }
    b.lead_guard = guard;
  }
  b.mapped = b.va;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  bool fresh = false;
  {
    auto it = g_pool.find(b.mapped_bytes);
    if (it != g_pool.end()) {
      b.handle = it->second;
      g_pool.erase(it);
    } else {
      GCHECK(hipMemCreate(&b.handle, b.mapped_bytes, &prop, 0));
      fresh = true;
    }
  }
  GCHECK(hipMemMap(b.mapped, b.mapped_bytes, 0, b.handle, 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  GCHECK(hipMemSetAccess(b.mapped, b.mapped_bytes, &acc, 1));
  if (fresh) settle(b.mapped, b.mapped_bytes);
  size_t slack = b.mapped_bytes - b.user_bytes;
  void* user = g_end_side ? (char*)b.mapped + slack : b.mapped;
  if (slack) fill(g_end_side ? b.mapped : (char*)b.mapped + b.user_bytes, kCanary, slack);
  // poison the tensor itself (0xFF.. = a NaN) so that reads of never-written memory show up (PG_GUARD_POISON=0: off)
  if (g_poison) fill(user, 0xFF, b.user_bytes);
  GCHECK(hipStreamSynchronize(nullptr));  // torch's side streams are non-blocking: the fills must have landed
  g_live[user] = b;
  return user;
}

__attribute__((visibility("default"))) void pg_guard_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_arena && (char*)ptr >= g_arena && (char*)ptr < g_arena + g_arena_bytes) return;  // capture arena: never recycled
  auto it = g_live.find(ptr);
  if (it == g_live.end()) {
    fprintf(stderr, "[pg_guard] free of an unknown pointer %p\n", ptr);
    return;
  }
  if (capturing(stream)) return;  // a graph may keep using it: stays mapped (and checked by pg_guard_check_all)
  GCHECK(hipSetDevice(device));
  GCHECK(hipDeviceSynchronize());
  Block b = it->second;
  check_block(b, ptr);
  g_live.erase(it);
  GCHECK(hipMemUnmap(b.mapped, b.mapped_bytes));
  g_pool.emplace(b.mapped_bytes, b.handle);  // recycled, never released (test runs are short-lived)
  GCHECK(hipMemAddressFree(b.va, b.va_bytes));
  if (b.lead_guard) GCHECK(hipMemAddressFree(b.lead_guard, g_gran));
}

// Verifies the canaries of every live allocation (after a device sync). Returns the violation count so far.
__attribute__((visibility("default"))) int pg_guard_check_all() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_gran) return 0;
  GCHECK(hipDeviceSynchronize());
  for (auto& kv : g_live)
    if (check_block(kv.second, kv.first)) {
      // re-arm so the same damage is reported once
      size_t slack = kv.second.mapped_bytes - kv.second.user_bytes;
      fill(g_end_side ? kv.second.mapped : (char*)kv.second.mapped + kv.second.user_bytes, kCanary, slack);
      GCHECK(hipStreamSynchronize(nullptr));
    }
  return g_violations;
}

__attribute__((visibility("default"))) int pg_guard_violations() { return g_violations; }
__attribute__((visibility("default"))) const char* pg_guard_report() { return g_report.c_str(); }
__attribute__((visibility("default"))) long pg_guard_live() { return (long)g_live.size(); }
__attribute__((visibility("default"))) long pg_guard_unguarded() { return g_unguarded; }
__attribute__((visibility("default"))) long pg_guard_settle_retries() { return g_settle_retries; }

}  // extern "C"
