"""Are the kernels of two builds of the library the same instruction streams?  For changes that add template parameters or instantiations next to a kernel that bounds the step
(round 5: the WARM / CHECK parameters of viterbi3_kernel, the S8_MIN_WAVES / S8_EXP=128 hooks of symbol8k_kernel): the existing kernels must come out of the compiler unchanged.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o old.s gr_dvbt_amd/csrc/dvbt_hip.hip      (on the commit before)
    hipcc ... -o new.s gr_dvbt_amd/csrc/dvbt_hip.hip                                                                              (on the commit after)
    python tools/isa_same.py old.s new.s
compares every function body of old.s with its namesake in new.s (local labels renumbered, comments and the directives that carry the mangled name dropped; template arguments that
new.s spells out and old.s did not -- defaulted ones -- are mapped with --strip, a regular expression removed from new.s's names, e.g. --strip 'ELi72ELb0' )."""
import re
import sys


def funcs(path, strip=None):
    s = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:', s, re.S | re.M):
        lines = [re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r'\s*;.*$', '', l.strip())) for l in m.group(2).split('\n')]
        name = re.sub(strip, '', m.group(1)) if strip else m.group(1)
        out[name] = [l for l in lines if l and not l.startswith((';', '.amdhsa_kernel', '.section'))]
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    strip = None
    if '--strip' in sys.argv:
        strip = sys.argv[sys.argv.index('--strip') + 1]
        args.remove(strip)
    a, b = funcs(args[0]), funcs(args[1], strip)
    same = 0
    for k, v in a.items():
        if k not in b:
            print("missing in the new build:", k[:120])
        elif b[k] == v:
            same += 1
        else:
            print("DIFFERENT:", k[:120], len(v), "->", len(b[k]), "lines")
    print(f"{same} of {len(a)} functions identical; {len([k for k in b if k not in a])} new")
    return 0 if same == len(a) else 1


if __name__ == "__main__":
    sys.exit(main())
