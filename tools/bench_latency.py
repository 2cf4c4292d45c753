import sys, os
sys.path.insert(0, os.getcwd())
os.environ["LIBZKP_SNARK_KEY_DIR"] = os.path.join(os.getcwd(), "tests", "golden")
import libzkp_amd as z
for t in ("range", "threshold", "consistency", "equality", "membership", "improvement"):
    z.benchmark_proof_generation_numeric(t, 2)
    r = z.benchmark_proof_generation_numeric(t, 20)
    print("%-12s avg %.2f ms  min %.2f  max %.2f" % (t, r["avg_time_ms"], r["min_time_ms"], r["max_time_ms"]))
