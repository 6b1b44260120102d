// The work-to-workgroup maps of the row kernels (helib_amd/csrc/work_map.h: xcd_remap_id, md_tile, md_work) compiled
// for the host: each must hand every work item to exactly one workgroup at every launch size.
// TEST INFRASTRUCTURE: built by tests/, never linked into the product library.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../helib_amd/csrc/work_map.h"

// 0 when workgroups 0 .. nwg-1 map onto 0 .. nwg-1 one to one
extern "C" int check_xcd_remap(unsigned nwg)
{
  std::vector<unsigned char> seen(nwg, 0);
  for (unsigned id = 0; id < nwg; id++) {
    const unsigned w = hx::xcd_remap_id(id, nwg);
    if (w >= nwg || seen[w])
      return 1;
    seen[w] = 1;
  }
  return 0;
}
// chunks of consecutive work items per XCD (what the remap is for): 0 when XCD x's items are one contiguous range
extern "C" int check_xcd_remap_contiguous(unsigned nwg)
{
  for (unsigned x = 0; x < 8; x++) {
    unsigned lo = ~0u, hi = 0, n = 0;
    for (unsigned id = x; id < nwg; id += 8) {
      const unsigned w = hx::xcd_remap_id(id, nwg);
      lo = w < lo ? w : lo;
      hi = w > hi ? w : hi;
      n++;
    }
    if (n && hi - lo + 1 != n)
      return 1;
  }
  return 0;
}
// the launch of ntt_moddown_apply*_kernel for (nkeep rows, npb elements): returns the grid size (8 per_xcd), or
// a negative code when some (row, element) is missed, visited twice or out of range; *padding = idle workgroups
extern "C" long check_md_work(unsigned nkeep, unsigned npb, unsigned* padding)
{
  const hx::MdTile T = hx::md_tile(nkeep, npb);
  const unsigned grid = 8u * T.per_xcd;
  std::vector<unsigned char> seen((size_t)nkeep * npb, 0);
  unsigned pad = 0;
  for (unsigned blk = 0; blk < grid; blk++) {
    const hx::MdWork w = hx::md_work(blk, nkeep, npb);
    if (!w.active) {
      pad++;
      continue;
    }
    if (w.ri >= nkeep || w.pb >= npb)
      return -1;
    unsigned char& s = seen[(size_t)w.ri * npb + w.pb];
    if (s)
      return -2;
    s = 1;
  }
  for (unsigned char s : seen)
    if (!s)
      return -3;
  // the rows that share one element's x / S streams are neighbours in their XCD's dispatch order
  if (padding)
    *padding = pad;
  return (long)grid;
}
