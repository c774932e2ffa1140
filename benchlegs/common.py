"""What every route and leg of bench.py shares: the byte models and peaks, the leg guard, the one-line emitter, the CPU sample.

bench.py assembles the contract line; the legs (benchlegs/legs_*.py) each RETURN the object they contribute — or raise, which the
guard records under `leg_errors` on the line, by name, and swallows: a leg must never cost the line."""
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_POINT = 15            # 2 B Z16 + 3 B RGB8 + 10 B packed record
PACK_BYTES_PER_POINT = 33            # a2 twin: 12 B vertex + 8 B texcoord + 3 B RGB8 + 10 B record
HBM_PEAK_GBS = 8000.0                # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
INFINITY_CACHE_BYTES = 256 << 20     # MI355X memory-side cache: a ring whose inputs fit it is not an HBM measurement
# A collective that never completes (the RCCL paths have not met a multi-GPU box yet) must end the run, not hang it: the process
# group's watchdog gives up after this long.
PG_TIMEOUT = datetime.timedelta(seconds=300)
POLICY = {0: "ieee", 1: "certified", 2: "certified+identityR", 3: "certified+noOverflow",
          4: "certified+identityR+noOverflow"}



def cpu_baseline(width, height, streams, budget_s):
    """The reference's `-m -t<N>` path restated (oracle/pcs_oracle_simd.c), timed on this host by a CHILD process
    (oracle/cpu_baseline.py) BEFORE any GPU leg: the OpenMP team is bound (OMP_PROC_BIND=close, OMP_PLACES=cores are in the
    child's environment when libgomp initialises), its buffers are first-touched by the team, no torch / HIP runtime
    threads run beside it, and `value` is the median over >= 30 passes at the best thread count (best / p10 / p90 beside
    it). Bracket A = the reference's own timed region (memset + pack, deprojection excluded, :291-293); bracket B adds the
    CPU deprojection, i.e. what the fused GPU kernel does."""
    import subprocess
    env = dict(os.environ)
    env["OMP_PROC_BIND"] = "close"
    env["OMP_PLACES"] = "cores"
    env.pop("OMP_NUM_THREADS", None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--width", str(width), "--height", str(height),
                        "--streams", str(streams), "--seconds", str(budget_s)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=max(600.0, 20 * budget_s))
    if r.returncode != 0:
        raise RuntimeError("cpu baseline child failed: " + r.stderr[-600:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


class Leg:
    """A leg of the line must never cost the line: an exception inside is recorded under `leg_errors` and swallowed."""

    def __init__(self, out, name):
        self.out, self.name = out, name

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and issubclass(et, Exception):
            self.out.setdefault("leg_errors", {})[self.name] = f"{et.__name__}: {ev}"[:300]
            try:
                import torch
                torch.cuda.synchronize()
            except Exception:       # noqa: BLE001
                pass
            return True
        return False


def flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:          # noqa: BLE001
        pass
    sys.stdout.flush()


def emit(out):
    """The contract's ONE JSON line — and the LAST line on stdout: what C libraries left in the C stdio buffer (RCCL prints a
    version banner with printf when a communicator is created; into a pipe it would otherwise be flushed at exit, after this line)
    goes out first."""
    flush_c_stdio()
    print(json.dumps(out), flush=True)



def run_leg(out, name, fn, *a, key=None, into=None, **kw):
    """Run one leg under the guard. fn returns the leg's object (stored as (into or out)[key or name]) or None (nothing to report: the
    leg does not apply). An exception is recorded as out["leg_errors"][name] — the leg's NAME stands on the line — and swallowed."""
    with Leg(out, name):
        r = fn(*a, **kw)
        if r is not None:
            (out if into is None else into)[key or name] = r
        return r
    return None


def frac_of_hbm(algo_bytes, ms):
    """(GB/s, fraction of the HBM peak) for algo_bytes moved in ms milliseconds."""
    gbs = algo_bytes / (ms * 1e-3) / 1e9
    return round(gbs, 1), round(gbs / HBM_PEAK_GBS, 4)
