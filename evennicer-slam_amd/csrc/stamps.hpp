// s_memtime stamps for the diagnostic builds (-DENS_STAMPS): per-wave cycle totals per code segment.
#pragma once
#ifdef ENS_STAMPS
#define ENS_NSEG 12
static __device__ unsigned long long* g_stamp_buf = nullptr;      // one per translation unit (each has its own setter)
#define STAMP_DECL unsigned long long st_acc[ENS_NSEG] = {}; unsigned long long st_prev = 0; unsigned long long st_rt0 = 0;
#define STAMP_START { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0)::"memory"); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#define STAMP(k) { unsigned long long st_now; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory"); __builtin_amdgcn_sched_barrier(0); st_acc[k] += st_now - st_prev; st_prev = st_now; }
#define STAMP_FLUSH { { unsigned long long rt1_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_)::"memory"); st_acc[ENS_NSEG - 1] = rt1_ - st_rt0; } if (g_stamp_buf && lane == 0) { for (int k_ = 0; k_ < ENS_NSEG; ++k_) g_stamp_buf[((size_t)blockIdx.x * 4 + wave) * ENS_NSEG + k_] = st_acc[k_]; } }
// absolute s_memrealtime (100 MHz) marks of a wave's lane 0: [workgroup][wave 0..7][16 slots] behind the per-segment totals
#define ENS_TL_BASE (2 * 256 * 4 * ENS_NSEG)
#define TL(slot) { if (g_stamp_buf && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); g_stamp_buf[ENS_TL_BASE + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = t_; } }
#else
#define STAMP_DECL
#define STAMP_START
#define STAMP(k)
#define STAMP_FLUSH
#define TL(slot)
#endif

// the same through a context object handed to device functions (forward kernel)
#ifdef ENS_STAMPS
struct StampCtx { unsigned long long acc[ENS_NSEG]; unsigned long long prev; };
#define FST_INIT(sx) { for (int k_ = 0; k_ < ENS_NSEG; ++k_) (sx).acc[k_] = 0; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"((sx).prev)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#define FST(sx, k) { unsigned long long n_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_)::"memory"); __builtin_amdgcn_sched_barrier(0); (sx).acc[k] += n_ - (sx).prev; (sx).prev = n_; }
#define FST_FLUSH(sx, wg, wave, lane) { if (g_stamp_buf && (lane) == 0) { for (int k_ = 0; k_ < ENS_NSEG; ++k_) g_stamp_buf[((size_t)(wg) * 4 + (wave)) * ENS_NSEG + k_] = (sx).acc[k_]; } }
#else
struct StampCtx {};
#define FST_INIT(sx)
#define FST(sx, k)
#define FST_FLUSH(sx, wg, wave, lane)
#endif
