// tools/emu/emu_race.cpp — a data-race detector for the emulated kernels (TEST INFRASTRUCTURE; tools/emu/README.md).
//
// `tools/emu/build.sh race` compiles the kernel source with -fsanitize=thread — for the INSTRUMENTATION only: every load and store becomes
// a call to __tsan_readN / __tsan_writeN — and links this file instead of the ThreadSanitizer runtime.  The hooks below know which
// workgroup, wave and barrier epoch of which launch the running lane belongs to (the emulator's scheduler tells them) and keep, per
// 4-byte granule, the last write and the last read of the CURRENT launch.  Reported, per pair of code addresses:
//   * two waves of one workgroup touching the same granule between the same two barriers, at least one writing, not both atomic;
//   * two workgroups of one launch touching the same global granule, at least one writing, not both atomic
//     (nothing orders the workgroups of a launch; LDS is private to a workgroup and exempt).
// Lanes of one wave are checked against each other only with EMU_RACE_LANES=1 (same wave, same barrier epoch, no wave collective passed by
// either since: an exchange that relies on the wave's lock step alone — legal on the device, but nothing the compiler promises).  One legal schedule is observed, but a conflict is reported whichever of
// the two accesses came first in it — unlike a digest mismatch it does not need the unlucky schedule to happen.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <link.h>
#include <dlfcn.h>
#include <map>
#include <string>
#include <vector>

namespace emu {
struct dim3_ { uint32_t x, y, z; };
struct LaneView { void* sp; dim3_ tid; uint32_t lane; };       // the head of emu::Lane (hip_runtime.h)
extern LaneView* cur_view() __attribute__((weak));
extern uint64_t race_launch, race_block;                          // set by the scheduler (emu_engine.inc)
extern uint32_t race_epoch;
extern int in_atomic;
uintptr_t cur_stack();
uint32_t cur_cep();
extern const size_t stack_bytes;
}  // namespace emu

namespace {
struct Acc { uint64_t launch; uint64_t block; const void* pc; uint32_t epoch, cep; uint16_t wave; uint8_t lane, atomic, valid; };
struct Cell { uintptr_t key; Acc w, r; };
constexpr size_t kCells = size_t(1) << 23;                        // 8 M granules of the current launch (stale launches are overwritten)
Cell* g_tab = nullptr;
uintptr_t g_img_lo = 0, g_img_hi = 0;                             // the library's own writable image: __shared__ statics and the dynamic-LDS buffer
uint64_t g_overflow = 0, g_checked = 0;
struct Rep { uint64_t n; uintptr_t addr; uint64_t b1, b2; uint32_t w1, w2; const char* kind; };
std::map<std::pair<const void*, const void*>, Rep>* g_reports = nullptr;
uintptr_t g_base = 0;
bool g_lanes_too = false;

int phdr_cb(struct dl_phdr_info* info, size_t, void* self) {
  Dl_info di;
  if (!dladdr(self, &di) || (uintptr_t)di.dli_fbase != info->dlpi_addr) return 0;
  g_base = info->dlpi_addr;
  for (int i = 0; i < info->dlpi_phnum; i++) {
    const ElfW(Phdr)& p = info->dlpi_phdr[i];
    if (p.p_type == PT_LOAD && (p.p_flags & PF_W)) {
      uintptr_t lo = info->dlpi_addr + p.p_vaddr, hi = lo + p.p_memsz;
      if (!g_img_lo || lo < g_img_lo) g_img_lo = lo;
      if (hi > g_img_hi) g_img_hi = hi;
    }
  }
  return 1;
}
void init() {
  if (g_tab) return;
  g_tab = (Cell*)calloc(kCells, sizeof(Cell));
  g_reports = new std::map<std::pair<const void*, const void*>, Rep>;
  dl_iterate_phdr(phdr_cb, (void*)&init);
  g_lanes_too = getenv("EMU_RACE_LANES") != nullptr;
}
void report(const Acc& old, const void* pc, uintptr_t addr, uint64_t block, uint32_t wave, const char* kind) {
  auto key = std::make_pair(old.pc, pc);
  auto it = g_reports->find(key);
  if (it == g_reports->end()) (*g_reports)[key] = Rep{1, addr, old.block, block, old.wave, wave, kind};
  else it->second.n++;
}
// 0 = none, 1 = two workgroups, 2 = two waves, 3 = two lanes of a wave (EMU_RACE_LANES=1 only)
inline int conflicts(const Acc& a, uint64_t launch, uint64_t block, uint32_t wave, uint32_t lane, uint32_t epoch, uint32_t cep, bool lds) {
  if (!a.valid || a.launch != launch) return 0;
  if (a.block != block) return lds ? 0 : 1;                        // another workgroup of the same launch
  if (a.epoch != epoch) return 0;                                  // a workgroup barrier in between
  if (a.wave != wave) return 2;                                    // another wave, no barrier in between
  return g_lanes_too && a.lane != lane && a.cep == cep ? 3 : 0;    // another lane of the wave, no collective in between: relies on lock step
}
void access(uintptr_t addr, size_t size, bool write, const void* pc) {
  emu::LaneView* l = emu::cur_view ? emu::cur_view() : nullptr;
  if (!l) return;                                                  // host code
  { const uintptr_t st = emu::cur_stack(); if (addr - st < emu::stack_bytes) return; }   // the lane's own fiber stack (locals): private by construction
  init();
  const uint64_t launch = emu::race_launch, block = emu::race_block;
  const uint32_t wave = (l->tid.x + 0u) / 64u + 16u * l->tid.y, lane = l->tid.x & 63u, epoch = emu::race_epoch, cep = emu::cur_cep();   // (every kernel of this library is 1-D: tid.x / 64)
  const bool atomic = emu::in_atomic != 0;
  for (uintptr_t g = addr >> 2; g <= (addr + size - 1) >> 2; g++) {
    const bool lds = (g << 2) >= g_img_lo && (g << 2) < g_img_hi;
    size_t h = (size_t)((g * 0x9E3779B97F4A7C15ull) >> 41) & (kCells - 1);
    Cell* c = nullptr;
    for (int probe = 0; probe < 64; probe++, h = (h + 1) & (kCells - 1)) {
      Cell& t = g_tab[h];
      const bool stale = (!t.w.valid || t.w.launch != launch) && (!t.r.valid || t.r.launch != launch);
      if (t.key == g && !stale) { c = &t; break; }
      if (stale) { memset(&t, 0, sizeof t); t.key = g; c = &t; break; }
    }
    if (!c) { g_overflow++; continue; }
    g_checked++;
    static const char* const kind_w[2][4] = {{"", "write/read, two workgroups", "write/read, two waves", "write/read, two lanes of a wave"},
                                             {"", "write/write, two workgroups", "write/write, two waves", "write/write, two lanes of a wave"}};
    if (!(atomic && c->w.atomic)) if (int k = conflicts(c->w, launch, block, wave, lane, epoch, cep, lds)) report(c->w, pc, g << 2, block, wave, kind_w[write][k]);
    if (write && !(atomic && c->r.atomic)) if (int k = conflicts(c->r, launch, block, wave, lane, epoch, cep, lds))
      report(c->r, pc, g << 2, block, wave, k == 1 ? "read/write, two workgroups" : k == 2 ? "read/write, two waves" : "read/write, two lanes of a wave");
    Acc a{launch, block, pc, epoch, cep, (uint16_t)wave, (uint8_t)lane, (uint8_t)atomic, 1};
    if (write) c->w = a; else c->r = a;
  }
}
struct AtExit {
  ~AtExit() {
    const char* out = getenv("EMU_RACE_OUT");
    if (!g_reports || !out) return;
    FILE* f = fopen(out, "a");
    if (!f) return;
    fprintf(f, "# %zu distinct pairs of code addresses; %llu granule accesses checked, %llu not tracked (shadow table full); library base %#zx\n",
            g_reports->size(), (unsigned long long)g_checked, (unsigned long long)g_overflow, (size_t)g_base);
    for (auto& kv : *g_reports)
      fprintf(f, "%#zx %#zx %llu %s | first: address %#zx workgroups %llu / %llu waves %u / %u\n", (size_t)((uintptr_t)kv.first.first - g_base), (size_t)((uintptr_t)kv.first.second - g_base),
              (unsigned long long)kv.second.n, kv.second.kind, (size_t)kv.second.addr, (unsigned long long)kv.second.b1, (unsigned long long)kv.second.b2, kv.second.w1, kv.second.w2);
    fclose(f);
  }
} g_at_exit;
}  // namespace

#define PC __builtin_extract_return_addr(__builtin_return_address(0))
extern "C" {
void __tsan_init() {}
void __tsan_func_entry(void*) {}
void __tsan_func_exit() {}
void __tsan_vptr_update(void**, void*) {}
void __tsan_vptr_read(void**) {}
void __tsan_read1(void* p) { access((uintptr_t)p, 1, false, PC); }
void __tsan_read2(void* p) { access((uintptr_t)p, 2, false, PC); }
void __tsan_read4(void* p) { access((uintptr_t)p, 4, false, PC); }
void __tsan_read8(void* p) { access((uintptr_t)p, 8, false, PC); }
void __tsan_read16(void* p) { access((uintptr_t)p, 16, false, PC); }
void __tsan_write1(void* p) { access((uintptr_t)p, 1, true, PC); }
void __tsan_write2(void* p) { access((uintptr_t)p, 2, true, PC); }
void __tsan_write4(void* p) { access((uintptr_t)p, 4, true, PC); }
void __tsan_write8(void* p) { access((uintptr_t)p, 8, true, PC); }
void __tsan_write16(void* p) { access((uintptr_t)p, 16, true, PC); }
void __tsan_unaligned_read2(void* p) { access((uintptr_t)p, 2, false, PC); }
void __tsan_unaligned_read4(void* p) { access((uintptr_t)p, 4, false, PC); }
void __tsan_unaligned_read8(void* p) { access((uintptr_t)p, 8, false, PC); }
void __tsan_unaligned_read16(void* p) { access((uintptr_t)p, 16, false, PC); }
void __tsan_unaligned_write2(void* p) { access((uintptr_t)p, 2, true, PC); }
void __tsan_unaligned_write4(void* p) { access((uintptr_t)p, 4, true, PC); }
void __tsan_unaligned_write8(void* p) { access((uintptr_t)p, 8, true, PC); }
void __tsan_unaligned_write16(void* p) { access((uintptr_t)p, 16, true, PC); }
void __tsan_read_range(void* p, unsigned long n) { if (n) access((uintptr_t)p, n, false, PC); }
void __tsan_write_range(void* p, unsigned long n) { if (n) access((uintptr_t)p, n, true, PC); }
void* __tsan_memcpy(void* d, const void* s, unsigned long n) { if (n) { access((uintptr_t)s, n, false, PC); access((uintptr_t)d, n, true, PC); } return memcpy(d, s, n); }
void* __tsan_memmove(void* d, const void* s, unsigned long n) { if (n) { access((uintptr_t)s, n, false, PC); access((uintptr_t)d, n, true, PC); } return memmove(d, s, n); }
void* __tsan_memset(void* d, int v, unsigned long n) { if (n) access((uintptr_t)d, n, true, PC); return memset(d, v, n); }
}
// (the host side's function-local statics are guarded by an acquire load, which the instrumentation turns into this)
extern "C" char __tsan_atomic8_load(const volatile char* p, int) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
