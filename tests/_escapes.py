"""Bookkeeping of the ONE tolerance the peak-bin bar has: a frame whose ORACLE top-2 magnitudes are within a
relative gap (1e-5; 1e-4 for the fp32 generator pipeline) may name either of the two bins.  Every test that accepts such a
frame records it here, and tests/test_zy_escape_hatches.py — collected after all parity tests — asserts that on the
committed seeds the hatch never fired, so a kernel change that starts to lean on it is visible."""
HITS = []          # (site, detail)
CHECKED = [0]      # peak bins compared so far in this process


def note(site, count, detail=""):
    if count:
        HITS.append((site, int(count), str(detail)))


def compared(n):
    CHECKED[0] += int(n)
