// Canary / guard-page device allocator for the GPU parity tests (TEST INFRASTRUCTURE, never loaded by the package).
//
// Plugged into torch with torch.cuda.memory.CUDAPluggableAllocator (tests/guard/__init__.py, PG_GUARD=1): every
// tensor of a test run then lives in its OWN device allocation between two margins filled with a canary byte,
//
//     [ 64 KB canary (0xFF) | the tensor (poisoned with 0xFF = NaN) | 64 KB canary (0xFF) ]
//
// so that an out-of-bounds WRITE of a kernel is no longer absorbed by the caching allocator's 2-20 MB segments: the
// margins are verified when the tensor is freed and, by the fixture tests/conftest.py installs, after every test
// (pg_guard_check_all); violations are counted and described with offsets relative to the tensor and the bytes that
// were written. The NaN poison makes reads of never-written memory visible in the tests' results.
//
// PG_GUARD_MODE=vmm adds guard PAGES (hipMemAddressReserve / hipMemCreate / hipMemMap: the tensor flush against an
// unmapped page so that out-of-bounds READS fault at the launch): on this runtime (ROCm 7.2) mappings made that way
// did not hold data reliably — canaries no kernel had touched read back as zeros or as another block's bytes, copy
// engines mishandled pointers into them — so it is opt-in, for experiments only; the margin mode below uses plain
// hipMalloc memory and kernels for every fill and check.
//
// hipGraph capture: nothing may be allocated, synchronised or freed while a stream captures, so allocations made
// during a capture come from a plain arena (no canaries) that is never recycled, and frees are deferred for ever —
// the eager warm-up iterations in front of every capture run the same kernels on guarded tensors.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

__global__ void guard_fill_kernel(unsigned char* p, unsigned char v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// res[0] = damaged bytes, res[1] = first damaged index, res[2] = last damaged index
__global__ void guard_check_kernel(const unsigned char* p, unsigned char v, size_t n, unsigned long long* res) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != v) {
      atomicAdd(&res[0], 1ull);
      atomicMin(&res[1], (unsigned long long)i);
      atomicMax(&res[2], (unsigned long long)i);
    }
}
// res[3..10] = the first eight damaged bytes, res[11..18] = the last eight
__global__ void guard_peek_kernel(const unsigned char* p, size_t first, size_t last, unsigned long long* res) {
  const int t = threadIdx.x;
  if (t < 8 && first + t <= last) res[3 + t] = p[first + t];
  if (t >= 8 && t < 16 && last >= (size_t)(15 - t)) res[3 + t] = p[last - (15 - t)];
}

struct Block {
  void* base;          // what hipMalloc returned (margin mode) / the reserved range (vmm mode)
  size_t base_bytes;
  void* mapped;        // vmm mode: first mapped byte
  size_t mapped_bytes;
  hipMemGenericAllocationHandle_t handle;
  bool vmm;
  size_t user_bytes;   // rounded size handed to torch
  size_t asked_bytes;  // size torch asked for
  size_t lead, trail;  // canary bytes in front of / behind the tensor
  unsigned long serial;
};

std::mutex g_mu;
std::unordered_map<void*, Block> g_live;
bool g_init = false, g_vmm = false, g_poison = true;
unsigned char g_poison_byte = 0xFF;  // PG_GUARD_POISON_BYTE: 0xFF = NaN (default); 0x5C = 2.5e17, a huge FINITE value that
                                     // survives the max / select operations (ReLU, masks) which swallow a NaN
size_t g_gran = 4096, g_align = 512, g_margin = 64 << 10;
int g_violations = 0;
unsigned long g_serial = 0;
std::string g_report;
char* g_arena = nullptr;
size_t g_arena_bytes = 0, g_arena_used = 0;
unsigned long long* g_res = nullptr;  // 19 words of plain device memory for the check kernels
const unsigned char kCanary = 0xFF;  // = the poison: a float read from a margin is a NaN, so an out-of-bounds READ whose value is used
                                      // (even multiplied by a zero weight) turns the test's result into NaN

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
  if (g_init) return;
  g_init = true;
  if (const char* s = getenv("PG_GUARD_MODE")) g_vmm = strcmp(s, "vmm") == 0;
  if (const char* a = getenv("PG_GUARD_ALIGN")) g_align = (size_t)atol(a);
  if (g_align < 16) g_align = 16;
  if (const char* s = getenv("PG_GUARD_MARGIN_KB")) g_margin = (size_t)atol(s) << 10;
  if (const char* s = getenv("PG_GUARD_POISON")) g_poison = atoi(s) != 0;
  if (const char* s = getenv("PG_GUARD_POISON_BYTE")) g_poison_byte = (unsigned char)strtol(s, nullptr, 0);
  if (g_vmm) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    GCHECK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
  }
  size_t arena_mb = 32768;  // captured steps allocate every activation and workspace of a training step, never recycled
  if (const char* s = getenv("PG_GUARD_ARENA_MB")) arena_mb = (size_t)atol(s);
  g_arena_bytes = arena_mb << 20;
  GCHECK(hipMalloc((void**)&g_arena, g_arena_bytes));
  GCHECK(hipMalloc((void**)&g_res, 19 * sizeof(unsigned long long)));
  fprintf(stderr, "[pg_guard] active: %s, size rounding %zu B, fresh tensors %s, capture arena %zu MB\n",
          g_vmm ? "guard pages (vmm, experimental)" : "canary margins of plain device memory", g_align,
          g_poison ? "poisoned with NaN bytes" : "left as allocated", arena_mb);
}

int g_capture_depth = 0;  // set from Python around every torch.cuda.graph(...) block (tests/guard/__init__.py): a free
                          // arrives with the stream its block was ALLOCATED on, which says nothing about a capture
                          // running on the current stream

bool capturing(hipStream_t stream) {
  if (g_capture_depth > 0) return true;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}

void fill(void* p, unsigned char v, size_t n) {
  if (!n) return;
  size_t blocks = (n + 256 * 64 - 1) / (256 * 64);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(guard_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, nullptr, (unsigned char*)p, v, n);
}

// one canary region of a block; `rel` = offset of its first byte relative to the tensor start. Returns damaged bytes.
size_t check_region(const Block& b, const unsigned char* from, size_t n, long rel) {
  if (!n) return 0;
  unsigned long long host[19] = {0, ~0ull, 0};
  GCHECK(hipMemcpy(g_res, host, sizeof host, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(guard_check_kernel, dim3(n >= 4096 ? 16 : 1), dim3(256), 0, nullptr, from, kCanary, n, g_res);
  GCHECK(hipMemcpy(host, g_res, 3 * sizeof host[0], hipMemcpyDeviceToHost));
  const size_t bad = (size_t)host[0];
  if (!bad) return 0;
  const size_t first = (size_t)host[1], last = (size_t)host[2];
  hipLaunchKernelGGL(guard_peek_kernel, dim3(1), dim3(64), 0, nullptr, from, first, last, g_res);
  GCHECK(hipMemcpy(host, g_res, sizeof host, hipMemcpyDeviceToHost));
  char line[400];
  int len = snprintf(line, sizeof line,
                     "allocation #%lu (%zu bytes asked, %zu given): %zu canary bytes overwritten, offsets %ld..%ld relative "
                     "to the tensor start (%s it); first bytes:",
                     b.serial, b.asked_bytes, b.user_bytes, bad, rel + (long)first, rel + (long)last,
                     rel < 0 ? "in front of" : "behind");
  for (int i = 0; i < 8 && first + i <= last; i++) len += snprintf(line + len, sizeof line - len, " %02llx", host[3 + i]);
  len += snprintf(line + len, sizeof line - len, " last bytes:");
  for (int i = 8; i < 16; i++)
    if (last >= (size_t)(15 - i)) len += snprintf(line + len, sizeof line - len, " %02llx", host[3 + i]);
  snprintf(line + len, sizeof line - len, "\n");
  g_report += line;
  fprintf(stderr, "[pg_guard] VIOLATION %s", line);
  g_violations++;
  return bad;
}

// Verifies both canary regions of one block (device must be idle). Returns the number of damaged bytes.
size_t check_block(const Block& b, void* user) {
  size_t bad = check_region(b, (unsigned char*)user - b.lead, b.lead, -(long)b.lead);
  bad += check_region(b, (unsigned char*)user + b.user_bytes, b.trail, (long)b.user_bytes);
  return bad;
}

void* alloc_margin(Block& b) {
  b.vmm = false;
  b.lead = b.trail = g_margin;
  b.base_bytes = b.user_bytes + 2 * g_margin;
  GCHECK(hipMalloc(&b.base, b.base_bytes));
  return (char*)b.base + g_margin;
}

// experimental: [mapped pages | unmapped guard page], the tensor flush against the guard page
void* alloc_vmm(Block& b, int device) {
  b.vmm = true;
  b.mapped_bytes = round_up(b.user_bytes, g_gran);
  b.base_bytes = b.mapped_bytes + g_gran;
  GCHECK(hipMemAddressReserve(&b.base, b.base_bytes, g_gran, nullptr, 0));
  b.mapped = b.base;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  GCHECK(hipMemCreate(&b.handle, b.mapped_bytes, &prop, 0));
  GCHECK(hipMemMap(b.mapped, b.mapped_bytes, 0, b.handle, 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  GCHECK(hipMemSetAccess(b.mapped, b.mapped_bytes, &acc, 1));
  b.lead = b.mapped_bytes - b.user_bytes;
  b.trail = 0;
  return (char*)b.mapped + b.lead;
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
  b.serial = ++g_serial;
  void* user = g_vmm ? alloc_vmm(b, device) : alloc_margin(b);
  fill((unsigned char*)user - b.lead, kCanary, b.lead);
  fill((unsigned char*)user + b.user_bytes, kCanary, b.trail);
  // poison the tensor itself (0xFF.. = a NaN) so that reads of never-written memory show up (PG_GUARD_POISON=0: off)
  if (g_poison) fill(user, g_poison_byte, b.user_bytes);
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
  if (capturing(stream)) return;  // a graph may keep using it: stays allocated (and checked by pg_guard_check_all)
  GCHECK(hipSetDevice(device));
  GCHECK(hipDeviceSynchronize());
  Block b = it->second;
  check_block(b, ptr);
  g_live.erase(it);
  if (b.vmm) {
    GCHECK(hipMemUnmap(b.mapped, b.mapped_bytes));
    GCHECK(hipMemRelease(b.handle));
    GCHECK(hipMemAddressFree(b.base, b.base_bytes));
  } else {
    GCHECK(hipFree(b.base));
  }
}

// Verifies the canaries of every live allocation (after a device sync). Returns the violation count so far.
__attribute__((visibility("default"))) int pg_guard_check_all() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_init) return 0;
  GCHECK(hipDeviceSynchronize());
  for (auto& kv : g_live)
    if (check_block(kv.second, kv.first)) {  // re-arm so the same damage is reported once
      fill((unsigned char*)kv.first - kv.second.lead, kCanary, kv.second.lead);
      fill((unsigned char*)kv.first + kv.second.user_bytes, kCanary, kv.second.trail);
      GCHECK(hipStreamSynchronize(nullptr));
    }
  return g_violations;
}

__attribute__((visibility("default"))) void pg_guard_capture(int begin) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_capture_depth += begin ? 1 : -1;
  if (g_capture_depth < 0) g_capture_depth = 0;
}
__attribute__((visibility("default"))) int pg_guard_violations() { return g_violations; }
__attribute__((visibility("default"))) const char* pg_guard_report() { return g_report.c_str(); }
__attribute__((visibility("default"))) long pg_guard_live() { return (long)g_live.size(); }

}  // extern "C"
