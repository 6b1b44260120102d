// work_map.h -- the work-to-workgroup maps of the row kernels as pure functions of (workgroup id, launch size):
// no HIP types, so tests/cpp/work_map_test.cpp compiles them for the host and checks that every map is a bijection
// onto its work items at every launch shape the benchmark legs use (the maps depend on the launch size,
// and the parity tests at batch 4 / 2 exercise other index arithmetic than the timed launches at batch 128 / 64).
#pragma once

#if defined(__HIPCC__)
#define HXW __host__ __device__ __forceinline__
#else
#define HXW inline
#endif

namespace hx {

// XCD-aware work mapping: hardware places workgroup id on XCD (id % 8) (observed, used for speed
// only).  Remap so that each XCD works on a contiguous chunk of the (row, batch) space, i.e. on a
// few primes only: their twiddle tables (2*N*16 B each) then stay resident in that XCD's 4 MiB
// L2 instead of all primes' tables cycling through every L2.  Bijective for any grid size.
HXW unsigned xcd_remap_id(unsigned id, unsigned nwg)
{
  const unsigned xcd = id & 7u, slot = id >> 3, qd = nwg >> 3, r = nwg & 7u;
  return (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + slot;
}

// Tile shape of the mod-down apply kernels (ntt_kernels.hip): g row groups of rg rows, 8/g XCDs per group
// each taking `chunk` (poly, batch) elements; per_xcd = workgroups launched per XCD (grid = 8 per_xcd).
struct MdTile {
  unsigned g, rg, chunk, per_xcd;
};
HXW MdTile md_tile(unsigned nkeep, unsigned npb)
{
  MdTile t;
  t.g = 1;
  while (t.g < 8 && (nkeep + t.g - 1) / t.g > 8)
    t.g *= 2;
  t.rg = (nkeep + t.g - 1) / t.g;
  const unsigned xpg = 8 / t.g;
  t.chunk = (npb + xpg - 1) / xpg;
  t.per_xcd = t.rg * t.chunk;
  return t;
}
// 2-D XCD-aware tiling.  Every kept row of one (poly, batch) element re-reads the same
// x and S streams, and every element of one row re-reads the same twiddle table.  XCD k (the
// hardware places workgroup id on XCD id % 8) owns a tile of `rg` rows x a chunk of the
// elements: its twiddle footprint is rg tables (<= 8 x 16N bytes, L2 resident), and the rg
// workgroups that share x/S are consecutive in its dispatch order, so they load them while
// the lines are still in that XCD's L2 -- x/S cross the fabric nkeep/rg times instead of nkeep.
// Workgroup `block` of the 8 per_xcd launched -> kept row ri < nkeep and element pb < npb; active = false for the
// padding workgroups of an uneven tile (they leave as a whole, before any barrier).
struct MdWork {
  bool active;
  unsigned ri, pb;
};
HXW MdWork md_work(unsigned block, unsigned nkeep, unsigned npb)
{
  const MdTile T = md_tile(nkeep, npb);
  const unsigned xcd = block & 7u, slot = block >> 3;
  const unsigned grp = xcd % T.g, part = xcd / T.g;
  const unsigned r0 = grp * T.rg, pb0 = part * T.chunk;
  const unsigned nr = r0 < nkeep ? (T.rg < nkeep - r0 ? T.rg : nkeep - r0) : 0u;
  const unsigned nloc = pb0 < npb ? (T.chunk < npb - pb0 ? T.chunk : npb - pb0) : 0u;
  MdWork w;
  w.active = slot < nr * nloc;
  w.ri = w.active ? r0 + slot % nr : 0u;
  w.pb = w.active ? pb0 + slot / nr : 0u;
  return w;
}

}  // namespace hx
