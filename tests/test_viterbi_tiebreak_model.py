"""The tie-break scheme of viterbi3_kernel (gr_dvbt_amd/csrc/k_viterbi3.hpp), as an abstract model on the CPU.

The kernel keeps the trellis in place: cell c holds state rotl6(c, u mod 6) at step u, the two cells of a butterfly differ in bit 5 - (u mod 6)
of the cell index, and the cell whose state has bit 5 set is the 'upper' one.  The reference lets the UPPER predecessor win a tie
(d_viterbi.c:508-521).  The kernel compares (metric, tie-break bits) as one integer and arms the tie-break bits of three steps at a time
(bits 6, 7, 8 below the metric; after steps 3, 6, 7 and 8 of every 8-step window), never clearing the bit a finished step leaves behind.
Claim checked here, on random metrics with many ties: at every compare the winner is the one the reference's rule picks, i.e. the bits armed for
later steps agree between the two candidates and a stale lower bit never decides."""
import numpy as np


def upper(c, u):
    return (c >> (5 - u % 6)) & 1


def arm(c, steps):
    """tie-break field for the given (up to three) coming steps: the earliest step in the lowest bit"""
    return sum(upper(c, u) << k for k, u in enumerate(steps))


def test_three_step_arming_reproduces_the_upper_wins_rule():
    rng = np.random.default_rng(11)
    cells = np.arange(64)
    for trial in range(200):
        metric = rng.integers(0, 3, 64)                              # few distinct values: ties everywhere
        tb = np.array([arm(c, (0, 1, 2)) for c in cells])            # armed at the window start (with the origin stamp)
        for u in range(64):                                          # eight windows
            w = u % 8
            bit = 5 - u % 6
            delta = rng.integers(-1, 2, 64)
            delta = np.where((cells >> bit) & 1, delta[cells ^ (1 << bit)], delta)      # one delta per butterfly
            new_metric, new_tb = metric.copy(), tb.copy()
            for c in cells:
                pc = c ^ (1 << bit)
                # own candidate keeps its output with +delta, the partner offers -delta (see the kernel's header)
                mx, my = metric[c] + delta[c], metric[pc] - delta[c]
                key_x, key_y = (mx << 3) | tb[c], (my << 3) | tb[pc]
                assert key_x != key_y                               # the armed bit of this step differs between the two cells
                take_x = key_x > key_y
                # the reference: strictly larger metric wins, a tie goes to the candidate that comes from the upper state
                want_x = mx > my or (mx == my and upper(c, u) == 1)
                assert take_x == want_x, (trial, u, c)
                new_metric[c], new_tb[c] = (mx, tb[c]) if take_x else (my, tb[pc])
            metric, tb = new_metric, new_tb
            # re-arming points of the kernel: after the window's 3rd step (steps 4..6), after the 6th (only the 7th: bits 7:6 hold the stamp of the
            # two oldest inputs from here on -- arbitrary data below the armed bit), after the 7th (the 8th; the stamp stays), after the 8th (the
            # next window's first three)
            if w == 2:
                tb = np.array([arm(c, (u + 1, u + 2, u + 3)) for c in cells])
            elif w == 5:
                tb = np.array([(upper(c, u + 1) << 2) | int(rng.integers(0, 4)) for c in cells])
            elif w == 6:
                tb = np.array([(upper(c, u + 1) << 2) | (int(tb[c]) & 3) for c in cells])
            elif w == 7:
                tb = np.array([arm(c, (u + 1, u + 2, u + 3)) for c in cells])
            metric = metric - metric.min()
