"""Host bookkeeping behind the reference's cache / metrics entry points (python_api.rs:110-164 registers clear_cache,
get_cache_stats, get_performance_metrics, prove_range_cached, prove_threshold_optimized).  No arithmetic here: the proofs
themselves come from the HIP library.  Mirrors /root/reference/src/utils/performance.rs:24-215 -- a TTL cache of 1000
entries / 3600 s with least-frequently-used eviction, keys salted per process, and the global operation / cache counters."""
import hashlib
import os
import threading
import time


class ProofCache:
    """performance.rs:24-103"""

    def __init__(self, max_size=1000, ttl_seconds=3600):
        self._lock = threading.Lock()
        self._entries = {}                 # key -> [data, created_at, access_count]
        self.max_size, self.ttl = max_size, float(ttl_seconds)

    def get(self, key):
        with self._lock:
            e = self._entries.get(key)
            if e is not None:
                if time.time() - e[1] < self.ttl:
                    e[2] += 1
                    METRICS.record_cache_hit()
                    return e[0]
                del self._entries[key]
        METRICS.record_cache_miss()
        return None

    def put(self, key, data):
        with self._lock:
            if len(self._entries) >= self.max_size:
                victim = min(self._entries, key=lambda k: self._entries[k][2])
                del self._entries[victim]
            self._entries[key] = [bytes(data), time.time(), 1]

    def clear(self):
        with self._lock:
            self._entries.clear()

    def size(self):
        with self._lock:
            return len(self._entries)


class PerformanceMetrics:
    """performance.rs:158-211"""

    def __init__(self):
        self._lock = threading.Lock()
        self.operation_counts, self.operation_times = {}, {}
        self.cache_hits = self.cache_misses = 0

    def record_operation(self, operation, seconds):
        with self._lock:
            self.operation_counts[operation] = self.operation_counts.get(operation, 0) + 1
            self.operation_times.setdefault(operation, []).append(seconds)

    def record_cache_hit(self):
        with self._lock:
            self.cache_hits += 1

    def record_cache_miss(self):
        with self._lock:
            self.cache_misses += 1

    def snapshot(self):
        with self._lock:
            return dict(self.operation_counts), {k: list(v) for k, v in self.operation_times.items()}, self.cache_hits, self.cache_misses


METRICS = PerformanceMetrics()
CACHE = ProofCache()
_SALT = os.urandom(32)                     # performance.rs:139-155: keys differ across processes


def generate_cache_key(operation, params):
    return "%s:%s" % (operation, hashlib.sha256(_SALT + operation.encode() + params).hexdigest())


AVG_KEYS = (("range_proof", "avg_range_proof_time_ms"), ("equality_proof", "avg_equality_proof_time_ms"),
            ("threshold_proof", "avg_threshold_proof_time_ms"), ("membership_proof", "avg_membership_proof_time_ms"),
            ("improvement_proof", "avg_improvement_proof_time_ms"), ("consistency_proof", "avg_consistency_proof_time_ms"))


def performance_metrics():
    """advanced/mod.rs:39-80 (averages truncated to whole milliseconds like Duration::as_millis)."""
    counts, times, hits, misses = METRICS.snapshot()
    total = hits + misses
    out = {"cache_hit_rate": (hits / total) if total else 0.0, "cache_size": float(CACHE.size()), "cache_hits": float(hits), "cache_misses": float(misses)}
    for op, key in AVG_KEYS:
        if op in times and times[op]:
            out[key] = float(int(sum(times[op]) / len(times[op]) * 1e3))
    for op, c in counts.items():
        out["%s_count" % op] = float(c)
    out["total_operations"] = float(sum(counts.values()))
    return out
