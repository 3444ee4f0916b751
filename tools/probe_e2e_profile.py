"""cProfile of bench.end_to_end_leg: where the host time of Rater.train goes (top cumulative entries)"""
import cProfile
import json
import pstats
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
pr = cProfile.Profile()
pr.enable()
out = bench.end_to_end_leg(B)
pr.disable()
print(json.dumps(out))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
