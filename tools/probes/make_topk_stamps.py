"""Refresh tools/probes/patches/topk_planes_kernel.h.patch and topk_inst_3.hip.patch against the current kernel.

The cycle stamps of the top-K sweep are kept as patches (the shipped sources hold no timing code); a patch stops
applying when the kernel around its hunks changes.  This script puts the stamps back by anchor text -- each anchor must
occur exactly once, or the script says which one does not -- into a scratch copy of csrc/, and writes the two patches
as the difference.  usage: python tools/probes/make_topk_stamps.py
"""
import os
import shutil
import subprocess
import tempfile

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "movie_recommender_system_amd", "csrc")
scratch = tempfile.mkdtemp(prefix="mirec_stamps_")
dst = os.path.join(scratch, "csrc")
shutil.copytree(src, dst)
p=dst+'/topk_planes_kernel.h'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1,(a[:70],s.count(a))
    s=s.replace(a,b)
rep("// BLOCKS: the filter comes as one list per 32 queries", """// Phase timing (-DMIREC_DEBUG_TIMING): cycle counts of the sections of the kernel, of every wavefront, summed into a
// per-translation-unit counter array; the F = 128 unit exports mirec_topk_debug_read
// (tools/probes/topk_phase_timing.py).  0 prologue, 1 the sweep of a stretch as every wavefront sees it (between its
// two barriers), 2 marked tiles worked, 3 compactions at stretch ends, 4 final stage, 5 marked tiles (a count),
// 6 wavefronts (a count), 7 the sweep inside the sweeping wavefronts, 8 of that the loop over the tiles alone.
// -DMIREC_DEBUG_TIMING_WORKED instead: the sections of a marked tile's work (tools/probes/topk_worked_tile.py):
// 0 operand loads (with a wait for them that the kernel does not otherwise have), 1 products, 2 liked bits,
// 3 selection, 4 re-offers; 5 and 6 as above.
#if defined(MIREC_DEBUG_TIMING) || defined(MIREC_DEBUG_TIMING_WORKED)
#define MIREC_TK_STAMPS 1
static __device__ unsigned long long mirec_topk_dbg[16];
#endif
#if defined(MIREC_DEBUG_TIMING)
#define TKS_T() __builtin_readcyclecounter()
#define TKS_ADD(i, v) dbg[i] += (v)
#else
#define TKS_T() 0ull
#define TKS_ADD(i, v)
#endif
#if defined(MIREC_DEBUG_TIMING_WORKED)
#define TKW_T() __builtin_readcyclecounter()
#define TKW_ADD(i, v) dbg[i] += (v)
#else
#define TKW_T() 0ull
#define TKW_ADD(i, v)
#endif

// BLOCKS: the filter comes as one list per 32 queries""")
rep("    if (h == 0) cnt[q] = 0;\n", """#ifdef MIREC_TK_STAMPS
    unsigned long long dbg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const unsigned long long tstart = TKS_T();
    if (h == 0) cnt[q] = 0;
""")
rep("""    auto work_tile = [&](int t) __attribute__((always_inline)) {
        const int i0 = t * BI;
""","""    auto work_tile = [&](int t) __attribute__((always_inline)) {
        const int i0 = t * BI;
        const unsigned long long w0 = TKW_T();
""")
rep("""        f32x16 acc = zero;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = tk_mma(yh[s], qh[s], acc);  // hi hi
""","""        f32x16 acc = zero;
#ifdef MIREC_DEBUG_TIMING_WORKED
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned long long w1 = TKW_T();
        TKW_ADD(0, w1 - w0);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = tk_mma(yh[s], qh[s], acc);  // hi hi
""")
rep("""        // liked bits of this lane's 16 items: bit tk_rowpos(i)
        unsigned liked = 0u;
""","""        const unsigned long long w2 = TKW_T();
        TKW_ADD(1, w2 - w1);
        // liked bits of this lane's 16 items: bit tk_rowpos(i)
        unsigned liked = 0u;
""")
rep("""        unsigned fail = 0u;
        // candidates are few:""","""        const unsigned long long w3 = TKW_T();
        TKW_ADD(2, w3 - w2);
        unsigned fail = 0u;
        // candidates are few:""")
rep("""        int offers = 0;
        while (__ballot(fail != 0u) != 0ull) {
""","""        const unsigned long long w4 = TKW_T();
        TKW_ADD(3, w4 - w3);
        int offers = 0;
        while (__ballot(fail != 0u) != 0ull) {
""")
rep("""                }
            }
        }
    };
    for (int seg_a = 0; seg_a < ntile;) {
""","""                }
            }
        }
        TKW_ADD(4, TKW_T() - w4);
    };
    TKS_ADD(0, TKS_T() - tstart);
    for (int seg_a = 0; seg_a < ntile;) {
        const unsigned long long s0 = TKS_T();
""")
rep("""            if (sweeper) {
                sweep_stretch(seg_a, seg_b);
""","""            if (sweeper) {
                const unsigned long long ss = TKS_T();
                sweep_stretch(seg_a, seg_b);
                TKS_ADD(7, TKS_T() - ss);
""")
rep("""        // the marked tiles in ascending order (the filter is walked forward only)
        if (marks != 0ull) {
            while (marks != 0ull) {
                const int b = __builtin_ctzll(marks);
                marks &= marks - 1ull;
                work_tile(seg_a + b);
            }
            // thresholds only rise at a compaction: what the stretch has added is counted in before the next
            compact();
        }
""","""        const unsigned long long s1 = TKS_T();
        TKS_ADD(1, s1 - s0);
        // the marked tiles in ascending order (the filter is walked forward only)
        if (marks != 0ull) {
            while (marks != 0ull) {
                const int b = __builtin_ctzll(marks);
                marks &= marks - 1ull;
                work_tile(seg_a + b);
                TKS_ADD(5, 1);
                TKW_ADD(5, 1);
            }
            const unsigned long long s2 = TKS_T();
            TKS_ADD(2, s2 - s1);
            // thresholds only rise at a compaction: what the stretch has added is counted in before the next
            compact();
            TKS_ADD(3, TKS_T() - s2);
        }
""")
rep("""        if (lane < 32) reinterpret_cast<unsigned*>(my_flags)[lane] = 0u;
        int t = a;""","""        if (lane < 32) reinterpret_cast<unsigned*>(my_flags)[lane] = 0u;
        const unsigned long long sl = TKS_T();
        int t = a;""")
rep("""        swept_tile(t, a, seen, q1, thr1, cm1, N{});
""","""        swept_tile(t, a, seen, q1, thr1, cm1, N{});
        TKS_ADD(8, TKS_T() - sl);
""")
rep("""    // ---- margin set -> exact scores -> the k best, for the 32 queries of the wavefront
""","""    const unsigned long long tfin = TKS_T();
    // ---- margin set -> exact scores -> the k best, for the 32 queries of the wavefront
""")
rep("""            out_scores[uq * k + c] = -FLT_MAX;
        }
    }
}
""","""            out_scores[uq * k + c] = -FLT_MAX;
        }
    }
#ifdef MIREC_TK_STAMPS
    TKS_ADD(4, TKS_T() - tfin);
    dbg[6] = 1;
    if (lane == 0)
        for (int i = 0; i < 10; ++i) atomicAdd(&mirec_topk_dbg[i], dbg[i]);
#endif
}
""")
open(p,'w').write(s)
p=dst+'/topk_inst_3.hip'
s=open(p).read()
s+="""
#ifdef MIREC_TK_STAMPS
extern "C" int mirec_topk_debug_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mirec::mirec_topk_dbg), sizeof(unsigned long long) * 16) != hipSuccess)
        return -3;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mirec::mirec_topk_dbg), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#endif
"""
open(p,'w').write(s)
for f in ('topk_planes_kernel.h', 'topk_inst_3.hip'):
    out = subprocess.run(['diff', '-u', '--label', 'a/' + f, '--label', 'b/' + f, os.path.join(src, f), os.path.join(dst, f)],
                         capture_output=True, text=True).stdout
    open(os.path.join(root, 'tools', 'probes', 'patches', f + '.patch'), 'w').write(out)
shutil.rmtree(scratch)
print("patches refreshed")
