#!/usr/bin/env python
"""Where the host spends the time to first audio: bench.py's own first-audio trials (same request, same streamer consumer) with
cProfile around the generate() call of the later trials.  Usage: python tools/ttfa_profile.py [bench.py arguments] 2> report.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

_orig = bench.first_audio_trials


def trials(fn, n_trials):
    if os.environ.get("TTFA_FIRST"):
        # the FIRST trial on its own (it is the slow one: 14 vs 10.9 ms at 1.5B), then a later one, side by side
        for tag in ("first", "third"):
            if tag == "third":
                _orig(fn, 1)
            pr = cProfile.Profile()

            def fn_p(st, pr=pr):
                pr.enable()
                try:
                    return fn(st)
                finally:
                    pr.disable()
            lat1 = _orig(fn_p, 1)
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
            print(f"[ttfa] {tag} trial {lat1} ms", file=sys.stderr)
            print(s.getvalue(), file=sys.stderr)
        return _orig(fn, n_trials)
    lat = _orig(fn, 2)                                   # warm
    pr = cProfile.Profile()

    def fn_prof(st):
        pr.enable()
        try:
            return fn(st)
        finally:
            pr.disable()
    lat_p = _orig(fn_prof, 4)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(f"[ttfa] unprofiled trials {lat} ms, profiled trials {lat_p} ms (4 calls in the stats below)", file=sys.stderr)
    print(s.getvalue(), file=sys.stderr)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
    print(s.getvalue(), file=sys.stderr)
    return lat + _orig(fn, max(1, n_trials - 2))


bench.first_audio_trials = trials
if __name__ == "__main__":
    bench.main()
