"""Stand-in for the braceexpand package: '{000..003}' ranges and '{a,b}' lists."""
import re


def braceexpand(pattern):
    pattern = str(pattern)
    m = re.search(r'\{([^{}]*)\}', pattern)
    if not m:
        yield pattern
        return
    body = m.group(1)
    rng = re.fullmatch(r'(-?\d+)\.\.(-?\d+)', body)
    if rng:
        lo, hi = rng.group(1), rng.group(2)
        width = len(lo) if len(lo) == len(hi) else 0
        alts = [str(v).zfill(width) for v in range(int(lo), int(hi) + 1)]
    else:
        alts = body.split(',')
    for alt in alts:
        yield from braceexpand(pattern[:m.start()] + alt + pattern[m.end():])
