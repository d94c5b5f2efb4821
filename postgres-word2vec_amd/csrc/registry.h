// registry.h -- the registry of live backends (processes holding pinned handles) per PHYSICAL GPU.  Plain C++17, no HIP:
// tests/test_registry.py compiles it on its own and drives it from several processes.
//
// one PostgreSQL backend = one process = one HIP context.  The processes cannot see each other through HIP, and two
// things must be decided per process: how many hardware queues the runtime may create (read ONCE, when the runtime starts) and how
// much of the chip a persistent scan takes while other backends are searching.  A small table in /dev/shm -- one slot per process:
// pid, the number of host-buffer searches it has in flight, the physical GPUs it holds handles on -- answers both without any
// configuration: a process that finds other live backends ON ITS GPU when it starts takes GPU_MAX_HW_QUEUES = 2
// (profiles/r05_backends.txt: 2 / 4 backends at six queues each are together SLOWER than one; at two queues each 4.9 / 6.9 M
// queries/s), one that starts alone takes six (four pipeline lanes + spares); a host-buffer search that starts while another
// backend of the same GPU is searching runs its scan on half of the CUs (option scan_share = 0, the default: auto).  One process per
// GPU (bench.py --gpus N, one cluster per GPU) are NOT neighbours.  Slots of processes that died are recognised by kill(pid, 0).
// Everything here is advisory: a registry that cannot be opened means "alone".
//   FREDDY_GPU_REGISTRY=0       no registry (every process believes it is alone)
//   FREDDY_GPU_REGISTRY_NAME    the shm object's name (default /freddy_gpu_backends2.<euid>)
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <mutex>

namespace freddy {
namespace registry {

// devs: the physical GPUs the process holds pinned handles on, one bit each (all ones: cannot tell -- a neighbour of everybody)
struct BackendSlot { std::atomic<int32_t> pid; std::atomic<int32_t> busy; std::atomic<uint64_t> devs; };
constexpr int kBackendSlots = 256;
static_assert(sizeof(BackendSlot) == 16 && std::atomic<int32_t>::is_always_lock_free && std::atomic<uint64_t>::is_always_lock_free, "slot layout");

struct State {
  BackendSlot* slots = nullptr;
  int my_slot = -1;
  int handles = 0;   // pinned handles of this process: it is a registered backend while this is > 0
  std::once_flag once;
  std::mutex mu;
};
inline State& state() { static State s; return s; }

inline bool pid_alive(int32_t pid) { return pid > 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM); }

// idx-th entry of a comma list of small non-negative integers; -1: no such entry / not a number (UUIDs)
inline int list_entry(const char* env, int idx) {
  const char* p = env;
  for (int i = 0;; ++i) {
    while (*p == ' ') ++p;
    if (*p < '0' || *p > '9') return -1;
    char* end = nullptr;
    const long v = strtol(p, &end, 10);
    while (*end == ' ') ++end;
    if (*end != ',' && *end != 0) return -1;
    if (i == idx) return (v >= 0 && v < 64) ? (int)v : -1;
    if (*end == 0) return -1;
    p = end + 1;
  }
}
// The physical ordinal behind HIP device index `dev`, as far as the environment tells BEFORE any HIP call: HIP_VISIBLE_DEVICES (or
// CUDA_VISIBLE_DEVICES) indexes into what ROCR_VISIBLE_DEVICES leaves.  -1 = cannot tell.
inline int physical_device(int dev) {
  if (dev < 0) return -1;
  int d = dev;
  const char* hip = getenv("HIP_VISIBLE_DEVICES");
  if (!hip || !*hip) hip = getenv("CUDA_VISIBLE_DEVICES");
  if (hip && *hip && (d = list_entry(hip, d)) < 0) return -1;
  const char* rocr = getenv("ROCR_VISIBLE_DEVICES");
  if (rocr && *rocr && (d = list_entry(rocr, d)) < 0) return -1;
  return d < 64 ? d : -1;
}
inline uint64_t device_bit(int dev) {
  const int p = physical_device(dev);
  return p < 0 ? ~0ull : 1ull << p;
}

inline void release() {
  State& st = state();
  if (st.slots && st.my_slot >= 0) {
    st.slots[st.my_slot].busy.store(0);
    st.slots[st.my_slot].devs.store(0);
    st.slots[st.my_slot].pid.store(0);
  }
}
inline void open() {
  const char* off = getenv("FREDDY_GPU_REGISTRY");
  if (off && *off && strtol(off, nullptr, 10) == 0) return;
  char name[128];
  const char* given = getenv("FREDDY_GPU_REGISTRY_NAME");
  if (given && *given) snprintf(name, sizeof name, "%s%s", given[0] == '/' ? "" : "/", given);
  else snprintf(name, sizeof name, "/freddy_gpu_backends2.%u", (unsigned)geteuid());   // (2: slots with the device mask)
  const int fd = shm_open(name, O_RDWR | O_CREAT, 0600);
  if (fd < 0) return;
  const size_t bytes = sizeof(BackendSlot) * kBackendSlots;
  if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return; }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return;
  state().slots = static_cast<BackendSlot*>(p);
  atexit(release);
}
inline void claim() {   // (mu held)
  State& st = state();
  if (!st.slots || st.my_slot >= 0) return;
  const int32_t me = (int32_t)getpid();
  for (int pass = 0; pass < 2 && st.my_slot < 0; ++pass)
    for (int i = 0; i < kBackendSlots && st.my_slot < 0; ++i) {
      int32_t cur = st.slots[i].pid.load();
      if (cur == me) { st.my_slot = i; break; }                      // (a forked child re-registers under its own pid below)
      if (cur != 0 && (pass == 0 || pid_alive(cur))) continue;       // first pass: free slots only; second: slots of dead processes
      if (st.slots[i].pid.compare_exchange_strong(cur, me)) { st.slots[i].busy.store(0); st.slots[i].devs.store(0); st.my_slot = i; }
    }
}

// A process is a registered backend while it holds pinned handles (+1 per pinned index on HIP device `device`, -1 when it is freed).
inline void handles(int delta, int device) {
  State& st = state();
  std::call_once(st.once, open);
  std::lock_guard<std::mutex> lock(st.mu);
  st.handles += delta;
  if (st.handles > 0) {
    claim();
    // (bits are only added while handles are held: a process that freed its handle on one of two GPUs stays that GPU's neighbour
    // until it has freed all of them)
    if (delta > 0 && st.slots && st.my_slot >= 0) st.slots[st.my_slot].devs.fetch_or(device_bit(device));
  } else if (st.slots && st.my_slot >= 0) { release(); st.my_slot = -1; }
}
// live backends other than this process with handles on the physical GPU behind HIP device `device` (-1: on any GPU): registered
// (searching = false) or inside a host-buffer search right now (searching = true)
inline int others(bool searching, int device) {
  State& st = state();
  std::call_once(st.once, open);
  if (!st.slots) return 0;
  const int32_t me = (int32_t)getpid();
  const uint64_t mine = device_bit(device);
  int n = 0;
  for (int i = 0; i < kBackendSlots; ++i) {
    const int32_t pid = st.slots[i].pid.load(std::memory_order_relaxed);
    if (pid == 0 || pid == me) continue;
    if (searching && st.slots[i].busy.load(std::memory_order_relaxed) <= 0) continue;
    if ((st.slots[i].devs.load(std::memory_order_relaxed) & mine) == 0) continue;   // (a backend of another GPU)
    if (pid_alive(pid)) ++n;
  }
  return n;
}
// this process enters (+1) / leaves (-1) a host-buffer search
inline void busy(int delta) {
  State& st = state();
  std::call_once(st.once, open);
  if (st.slots && st.my_slot >= 0 && st.slots[st.my_slot].pid.load(std::memory_order_relaxed) == (int32_t)getpid())
    st.slots[st.my_slot].busy.fetch_add(delta);
}

}  // namespace registry
}  // namespace freddy
