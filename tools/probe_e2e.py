"""Diagnostic: where the wall time of bench.py's end-to-end leg (Rater.train over synthetic files) goes."""
import cProfile
import pstats
import sys

sys.path.insert(0, '.')
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
pr = cProfile.Profile()
pr.enable()
out = bench.end_to_end_leg(B)
pr.disable()
print(out)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
