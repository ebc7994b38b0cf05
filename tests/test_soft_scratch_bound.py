"""The decision scratch of a soft-decision handle (k_soft4.hpp): s4_grid is NOT monotone in the stream's length (s4_plan raises the chunk size B with it,
so the task count drops each time B increments), yet one handle launches segments of every length up to max_samples -- a stream's last piece
is always shorter.  The host functions are compiled for the CPU as they stand in the header: s4_grid_bound(max) must cover s4_grid(t) for every t <= max."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MAIN = r"""
#include <cstdio>
#include <cstddef>
namespace dvbt {
#include "s4_host.inc"
}
using namespace dvbt;
int main()
{
  // the example of the finding: one byte more, 31 workgroups fewer
  if (!(s4_grid(64ll * 32768, 24) == 2048 && s4_grid(64ll * 32768 + 1, 24) < 2048)) { printf("s4_grid is monotone after all?\n"); return 1; }
  long long bad = 0, checked = 0;
  const int ntbs[] = {5, 9, 10, 15, 24};
  // every length up to 6 M bytes in steps that hit every B transition region densely enough, plus a fine sweep around the powers of the round size
  for (int ni = 0; ni < 5; ni++) {
    unsigned run_max = 0;
    for (long long t = 1; t <= 6000000; t += (t < 70000 ? 1 : 97)) {
      const unsigned g = s4_grid(t, ntbs[ni]);
      if (g > run_max) run_max = g;
      checked++;
      if (run_max > s4_grid_bound(t)) bad++;       // some t' <= t needs more slots than a handle sized for t has
      if (s4_grid_bound(t) > (unsigned)S4_GRID) bad++;
    }
  }
  printf("%lld violations of %lld\n", bad, checked);
  return bad != 0;
}
"""


def test_scratch_bound_covers_every_shorter_segment(tmp_path):
    src = open(os.path.join(ROOT, "gr_dvbt_amd", "csrc", "k_soft4.hpp")).read()
    a, b = src.index("constexpr int S4_WARM"), src.index("struct S4Lane")
    (tmp_path / "s4_host.inc").write_text(src[a:b])
    (tmp_path / "main.cpp").write_text(MAIN)
    exe = str(tmp_path / "s4_host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", str(tmp_path), "-o", exe, str(tmp_path / "main.cpp")])
    out = subprocess.check_output([exe], text=True)
    assert out.startswith("0 violations"), out
