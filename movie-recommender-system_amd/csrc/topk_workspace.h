// Layout of the optional workspace of mirec_topk (host side only).
//
//   [0, 256)            header: u32 bits of max ||y|| at byte 0, i32 length of the query list at byte 4
//   [256, 256 + Lq)     query list: int32 per query (Lq = 4*nq rounded up to 256)
//   [.., .. + Ln)       largest row norm of every item tile: float per tile (Ln rounded up to 256)
//   [.., .. + Lo)       a sweep order made inside the call (no filter, no order given): 4096 bin counters, then
//                       int32 per item
//   [.., .. + Lp)       item factors split into two bf16 planes (hi, mid), tile by tile, three sections a tile,
//                       rows past the last item zero:
//                         hi  [32 items][f] bf16                  what the sweep copies into LDS row by row
//                         hi  [f/8 chunks][32 items][8] bf16      what a lane of a marked tile's products reads: chunk
//                         mid [f/8 chunks][32 items][8] bf16      2s+h of item c -- a wavefront's load is 1 KiB in a row
//   [.., .. + Ls)       per workgroup in flight: where half of its wavefronts leave the hi planes of their queries for
//                       the other half (one flag per slot, then the slots of 128 query rows of f bf16)
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace mirec {

constexpr int TOPK_PLANE_KMAX = 24;  // largest k the bf16-plane kernel serves
constexpr int TOPK_NPLANES = 2;      // bf16 planes per factor matrix
constexpr int TOPK_TILE_SECTIONS = 3;  // images of an item tile in the workspace
constexpr int TOPK_ORDER_BINS = 4096;  // bins of the in-call sweep order: the top 12 bits of a row norm
constexpr int TOPK_WG_QUERIES = 256;   // queries of a workgroup of the bf16-plane kernel
constexpr int TOPK_STASH_SLOTS = 1024; // several times the workgroups a device runs at once (one per CU)

inline bool topk_planes_supported(int f, int k) {
    return f >= 16 && f <= 128 && f % 16 == 0 && k >= 1 && k <= TOPK_PLANE_KMAX;
}

struct TopkWorkspace {
    size_t qlist_off, tnorm_off, bins_off, order_off, planes_off, flags_off, stash_off, bytes;
    int stash_slots;
};

inline TopkWorkspace topk_workspace_layout(int64_t nq, int64_t ni, int f) {
    TopkWorkspace w;
    w.qlist_off = 256;
    const size_t lq = ((size_t)nq * 4 + 255) / 256 * 256;
    const size_t ntile = (size_t)((ni + 31) / 32);
    w.tnorm_off = w.qlist_off + lq;
    w.bins_off = w.tnorm_off + (ntile * 4 + 255) / 256 * 256;
    w.order_off = w.bins_off + TOPK_ORDER_BINS * 4;
    w.planes_off = w.order_off + ((size_t)ni * 4 + 255) / 256 * 256;
    w.flags_off = w.planes_off + ntile * TOPK_TILE_SECTIONS * 32 * (size_t)f * 2;
    const int64_t blocks = (nq + TOPK_WG_QUERIES - 1) / TOPK_WG_QUERIES;
    w.stash_slots = (int)(blocks < TOPK_STASH_SLOTS ? (blocks > 0 ? blocks : 1) : TOPK_STASH_SLOTS);
    w.stash_off = w.flags_off + ((size_t)w.stash_slots * 4 + 255) / 256 * 256;
    w.bytes = w.stash_off + (size_t)w.stash_slots * (TOPK_WG_QUERIES / 2) * (size_t)f * 2;
    return w;
}

}  // namespace mirec
