// One-shot record exchange between the ranks of a node (include/espm_mu.h, espm_xchg_*; SURVEY.md 8b espm_allreduce, 8e).
//
// Per iteration every rank owns ONE record of a few tens of KB (its partial R H^T, the statistics and the boundary rows of
// its new H block) that every other rank needs before the W update: an all-gather of latency-bound messages on a strictly
// serial dependency chain.  A ring or tree collective is the wrong shape for that; here every rank WRITES its record into a
// slot of every peer's mailbox over the direct xGMI links (peer memory mapped through hipIpc) and raises a flag there; a
// rank waits until the flags of all ranks show the sequence number.  The fixed rank order of the slots makes the sum that
// follows bit-identical on every rank.
//
//   mailbox (uncached device memory, one per rank, mapped into every peer):
//     records [2][world][record_bytes]   parity = seq & 1: a record stays readable while the next exchange fills the other set
//     flags   [world] x 64 bytes         flag[r] = last sequence number rank r has delivered here
//     errors  1 x uint32                 waits that gave up (bounded spin: a lost peer must not hang the device)
//     wgflags [world][k n_pad / 32 + 1]  one flag per reduction workgroup of a rank (+ its extra workgroup): the in-launch exchange
//     gran    [2][world][34 x ...]       8-byte granules {value, sequence number} of the in-launch exchange: a reduction workgroup's 32
//                                        entries of A and the two halves of its row sum of the new H - value and "it is there" in ONE
//                                        store, so a piece costs one trip over the link instead of data, drain, flag
//   post(seq): one workgroup per destination copies the staged record into slot [seq & 1][rank] of that destination with
//              write-through system-scope stores; every thread drains its stores (s_waitcnt vmcnt(0)), the workgroup's barrier,
//              then ONE lane stores the flag (relaxed, system scope).
//
// ORDERING CONTRACT (what makes "flag seen => record there" hold; DESIGN.md section 5 has the long form).
//   producer: (1) every byte of a record is stored write-through at system scope (sc0 sc1: the store is acknowledged by the memory
//             it targets, not by a cache on the way - the mailboxes are uncached / fine-grained memory besides); (2) every storing wave
//             waits for the acknowledgement of ALL its stores (s_waitcnt vmcnt(0)); (3) a workgroup barrier collects the waves;
//             (4) one lane stores the flag.  This is MI355X_MICROARCH.md's form "sc0 sc1 stores and loads both sides" with its
//             conditions (2) and (3) of the consumer bullet ("every storing wave ran s_waitcnt vmcnt(0) after its stores, and each
//             flag store comes after the wait of EVERY wave it signals for: a lane that signals for other waves does so behind a
//             workgroup barrier"), which that guide measured INSIDE one device and calls "not an architectural guarantee".
//   consumer: polls the flag with relaxed system-scope loads (sc0 sc1: never served by its L1 / L2); the kernels that read the
//             records start behind the wait on the stream (a kernel boundary), and the in-launch exchange reads granules - 8-byte
//             {value, sequence number} words whose own arrival is the signal: no ordering between two stores is relied on there.
//   What is NOT established on this build's hardware (one GPU): that the acknowledgement of a write-through store to a PEER's
//   memory over xGMI implies its visibility to that peer's loads before a later store from the same wave becomes visible.  Two
//   guards: the start-up self-test (espm_amd/sharding.py: patterns that change with every sequence number; a record that arrives
//   after its flag shows as `corrupt`) runs under BOTH orders on whatever peers a run has, and ESPM_XCHG_ORDER=release /
//   espm_xchg_set_order(x, 1) replaces step (4) by a release store at system scope (the compiler's recipe: write back the L2,
//   wait, store) for a node on which the relaxed form shows a single corrupt record.
//   (espm_mu_shard_exchange_finish, mu_w_step.hip, does all of this INSIDE the slab-reduction launch, piece by piece, with
//    one flag per reduction workgroup: wgflags)
//   wait(seq): one workgroup, lane r polls flag[r] (system-scope loads, s_sleep between polls) until it reaches seq or
//              ~2 s have passed; the kernels that read the records are launched behind it on the same stream.
// Why the records of sequence s are safe to read until s + 2 is posted: a peer posts s + 2 only after its wait(s + 1)
// returned, i.e. after this rank posted s + 1, which it does (stream order) after everything that read the records of s.
#include <stdlib.h>
#include <string.h>

#include "mu_common.hpp"
#include "mu_xchg.hpp"

namespace espm {

constexpr int XCHG_MAX_WORLD = 16;
constexpr int XCHG_FLAG_STRIDE = 64;

struct XchgPostArgs {
  unsigned char* dst[XCHG_MAX_WORLD];   // peers' mailboxes
  const unsigned char* src;             // staged record
  size_t record_bytes, slot_off, flag_off;
  unsigned int seq;
  int release;
};

__global__ __launch_bounds__(256) void xchg_post_kernel(const XchgPostArgs a) {
  unsigned char* mb = a.dst[blockIdx.x];
  const unsigned long long* s = reinterpret_cast<const unsigned long long*>(a.src);
  unsigned long long* d = reinterpret_cast<unsigned long long*>(mb + a.slot_off);
  const size_t n8 = a.record_bytes / 8;
  // system-scope write-through stores: whatever memory type the peer's mapping has here, the data is on its way to that
  // rank's memory when the store is acknowledged
  for (size_t i = threadIdx.x; i < n8; i += 256) __hip_atomic_store(d + i, s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // what is needed then is ORDER (every thread's stores before the flag), not a system-scope fence, which would write back
  // this XCD's whole L2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int* flag = reinterpret_cast<unsigned int*>(mb + a.flag_off);
    if (a.release) __hip_atomic_store(flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(64) void xchg_wait_kernel(unsigned char* mailbox, size_t off_flags, size_t off_err, int world,
                                                       unsigned int seq, long long max_ticks) {
  const int r = threadIdx.x;
  if (r < world) {
    const unsigned int* flag = reinterpret_cast<const unsigned int*>(mailbox + off_flags + (size_t)r * XCHG_FLAG_STRIDE);
    const long long t0 = wall_clock64();
    // (sequence numbers only grow: >= also accepts a peer that is already one exchange ahead)
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (wall_clock64() - t0 > max_ticks) {   // a peer that never delivers must not hang the device: give up, count it
        atomicAdd(reinterpret_cast<unsigned int*>(mailbox + off_err), 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  // (the kernels that read the records start behind this one on the stream: a kernel boundary, and the mailbox is uncached)
}

}  // namespace espm

using namespace espm;

extern "C" {

int espm_xchg_create(int world, int rank, size_t record_bytes, espm_xchg** out) {
  ESPM_REQUIRE(out && world >= 1 && world <= XCHG_MAX_WORLD && rank >= 0 && rank < world, "xchg_create: world=%d rank=%d (at most %d ranks)", world,
               rank, XCHG_MAX_WORLD);
  ESPM_REQUIRE(record_bytes >= 16 && record_bytes % 16 == 0, "xchg_create: record_bytes=%zu must be a positive multiple of 16", record_bytes);
  espm_xchg* x = new espm_xchg();
  x->world = world;
  x->rank = rank;
  x->record_bytes = record_bytes;
  x->off_flags = 2 * (size_t)world * record_bytes;
  x->off_err = x->off_flags + (size_t)world * XCHG_FLAG_STRIDE;
  x->off_wgflags = x->off_err + XCHG_FLAG_STRIDE;
  x->wgflags = (int)(record_bytes / 128) + 2;   // >= k * ceil(n_pad / 32) + 1 for any record that holds k * n_pad floats
  x->off_gran = (x->off_wgflags + (size_t)world * x->wgflags * sizeof(uint32_t) + 63) / 64 * 64;
  x->mailbox_bytes = x->off_gran + 2 * (size_t)world * ((size_t)34 * x->wgflags + 2 * ESPM_HS_STRIDE) * sizeof(uint64_t);
  x->mailbox_bytes = (x->mailbox_bytes + 255) / 256 * 256;
  for (int r = 0; r < XCHG_MAX_WORLD; ++r) {
    x->peers[r] = nullptr;
    x->opened[r] = false;
  }
  // uncached: written by peers over the fabric and read here within the same launch sequence - no stale L2 lines
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&x->mailbox), x->mailbox_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) e = hipExtMallocWithFlags(reinterpret_cast<void**>(&x->mailbox), x->mailbox_bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    delete x;
    return check_hip(e, "xchg_create: mailbox allocation");
  }
  e = hipMalloc(reinterpret_cast<void**>(&x->staging), record_bytes);
  if (e == hipSuccess) e = hipMemset(x->mailbox, 0, x->mailbox_bytes);
  if (e == hipSuccess) e = hipMemset(x->staging, 0, record_bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(x->mailbox);
    if (x->staging) (void)hipFree(x->staging);
    delete x;
    return check_hip(e, "xchg_create: staging allocation");
  }
  x->peers[rank] = x->mailbox;
  const char* order = getenv("ESPM_XCHG_ORDER");
  x->order = (order && strcmp(order, "release") == 0) ? 1 : 0;
  *out = x;
  return ESPM_OK;
}

int espm_xchg_set_order(espm_xchg* x, int release) {
  ESPM_REQUIRE(x && (release == 0 || release == 1), "xchg_set_order: context and 0 (relaxed flag behind the drain) or 1 (release flag)");
  x->order = release;
  return ESPM_OK;
}

int espm_xchg_order(const espm_xchg* x) { return x ? x->order : -1; }

int espm_xchg_handle(const espm_xchg* x, void* handle_out) {
  ESPM_REQUIRE(x && handle_out, "xchg_handle: NULL pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == ESPM_XCHG_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h;
  if (int rc = check_hip(hipIpcGetMemHandle(&h, x->mailbox), "xchg_handle: hipIpcGetMemHandle")) return rc;
  memcpy(handle_out, &h, sizeof(h));
  return ESPM_OK;
}

int espm_xchg_connect(espm_xchg* x, const void* handles) {
  ESPM_REQUIRE(x && handles, "xchg_connect: NULL pointer");
  for (int r = 0; r < x->world; ++r) {
    if (r == x->rank || x->peers[r]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const unsigned char*>(handles) + (size_t)r * ESPM_XCHG_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    if (int rc = check_hip(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "xchg_connect: hipIpcOpenMemHandle")) return rc;
    x->peers[r] = static_cast<unsigned char*>(p);
    x->opened[r] = true;
  }
  return ESPM_OK;
}

void* espm_xchg_staging(const espm_xchg* x) { return x ? x->staging : nullptr; }

const void* espm_xchg_records(const espm_xchg* x, int parity) {
  if (!x) return nullptr;
  return x->mailbox + (size_t)(parity & 1) * x->world * x->record_bytes;
}

int espm_xchg_post(espm_xchg* x, uint32_t seq, espm_stream_t stream) {
  ESPM_REQUIRE(x, "xchg_post: NULL context");
  XchgPostArgs a;
  for (int r = 0; r < XCHG_MAX_WORLD; ++r) a.dst[r] = r < x->world ? x->peers[r] : nullptr;
  for (int r = 0; r < x->world; ++r) ESPM_REQUIRE(a.dst[r], "xchg_post: rank %d is not connected (espm_xchg_connect)", r);
  a.src = x->staging;
  a.record_bytes = x->record_bytes;
  a.slot_off = ((size_t)(seq & 1u) * x->world + x->rank) * x->record_bytes;
  a.flag_off = x->off_flags + (size_t)x->rank * XCHG_FLAG_STRIDE;
  a.seq = seq;
  a.release = x->order;
  hipLaunchKernelGGL(xchg_post_kernel, dim3(x->world), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_hip(hipGetLastError(), "xchg_post launch");
}

int espm_xchg_wait(espm_xchg* x, uint32_t seq, espm_stream_t stream) {
  ESPM_REQUIRE(x, "xchg_wait: NULL context");
  const long long max_ticks = 200000000LL;   // 2 s of the 100 MHz wall clock
  hipLaunchKernelGGL(xchg_wait_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), x->mailbox, x->off_flags, x->off_err,
                     x->world, seq, max_ticks);
  return check_hip(hipGetLastError(), "xchg_wait launch");
}

int espm_xchg_timeouts(const espm_xchg* x, uint32_t* count_out) {
  ESPM_REQUIRE(x && count_out, "xchg_timeouts: NULL pointer");
  return check_hip(hipMemcpy(count_out, x->mailbox + x->off_err, sizeof(uint32_t), hipMemcpyDeviceToHost), "xchg_timeouts");
}

int espm_xchg_destroy(espm_xchg* x) {
  if (!x) return ESPM_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < x->world; ++r)
    if (x->opened[r]) (void)hipIpcCloseMemHandle(x->peers[r]);
  (void)hipFree(x->mailbox);
  (void)hipFree(x->staging);
  delete x;
  return ESPM_OK;
}

}  // extern "C"
