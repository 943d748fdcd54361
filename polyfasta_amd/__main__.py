import os
import sys
import time

_t_enter = time.time()          # wall clock: bench.py's cold_start leg subtracts its own spawn time

# numpy's import starts one OpenBLAS thread per CPU the machine SHOWS, and they spin for a while: 256 threads and 1.7-2.3 CPU-seconds on
# a GPU box whose control group grants 16 CPUs — the whole quota of a scheduler period gone before the first file is read (the group
# is then frozen until the period ends: ~100 ms at some random later point, usually the exit), and 50 ms of import time.  Nothing
# here uses BLAS (profiles/r04_startup_cpu_cost.txt).  A value the user exported is kept.
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")


def _launcher_only(argv):
    """`--gpus N` with N > 1 and no launcher's environment: this process only starts the ranks (dist.spawn_ranks) and
    must not touch the GPU itself."""
    # argparse accepts any unambiguous prefix of a long option (`--gpu 4`, `--g=4`: no other option starts with --g) and, when
    # an option is repeated, its last occurrence
    last = None
    for i, a in enumerate(argv):
        if a == "--":
            break
        name, eq, val = a.partition("=")
        if len(name) >= 3 and "--gpus".startswith(name):
            last = val if eq else (argv[i + 1] if i + 1 < len(argv) else None)
    try:
        return last is not None and int(last) > 1
    except ValueError:
        return False


# A one-file run is dominated by start-up (interpreter, numpy, hipInit: bench.py `cold_start`).  The
# HIP runtime is brought up by a native thread WHILE Python imports numpy and parses the command line;
# pfa_init() joins it.  Ranks of a multi-GPU job bind their CPUs first and initialise in pfa_init();
# the launcher of `--gpus N` never initialises the GPU.
_prewarmed = None
if "WORLD_SIZE" not in os.environ and not os.environ.get("PFA_NO_PREWARM") and not _launcher_only(sys.argv[1:]):
    try:
        import ctypes
        _lib_path = os.environ.get("PFA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpfa_hip.so")
        _prewarmed = ctypes.CDLL(_lib_path)
        _prewarmed.pfa_prewarm(int(os.environ.get("LOCAL_RANK", "0")))
    except (OSError, AttributeError, ValueError):
        _prewarmed = None        # cli.main() reports a missing library in its own words

from . import cli  # noqa: E402

cli.startup_marks.append(("interpreter+package+prewarm start", _t_enter))
cli.startup_marks.append(("import cli (numpy, ctypes, argparse)", time.time()))
rc = 1
try:
    rc = cli.main()
except SystemExit as e:         # argparse errors, --help
    rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    if e.code is not None and not isinstance(e.code, int):
        print(e.code, file=sys.stderr)
except KeyboardInterrupt:
    rc = 130
except BaseException:           # noqa: BLE001  the reference's way out of anything else: a traceback, exit code 1
    import traceback
    traceback.print_exc()
    rc = 1
finally:
    # The prewarm thread may still be inside hipInit when main() ends before its first scan (closed stdin at the prompt,
    # an unreadable directory, Ctrl-C): never leave the process under it — pfa_shutdown() joins it (and, on these paths
    # only, tears the runtime down in order).  After a normal run pfa_init() has joined it long ago.
    # A run that ends well before any pfa_init (--help) only joins it.
    if _prewarmed is not None:
        try:
            if rc != 0:
                _prewarmed.pfa_shutdown()
            else:
                _prewarmed.pfa_prewarm_join()
        except Exception:       # noqa: BLE001
            pass
# every row has been written: leave without the interpreter's and the HIP runtime's teardown (45 ms of a
# 0.3 s one-file run); the driver releases the device when the process is gone
if os.environ.get("PFA_EXIT") in ("shutdown", "reset"):      # timing experiments: the library's own teardown first (buffers freed, stream destroyed), + hipDeviceReset
    _t = time.time()
    from . import _lib as _l
    _l.shutdown()
    if os.environ["PFA_EXIT"] == "reset":
        import ctypes as _c
        _c.CDLL("libamdhip64.so").hipDeviceReset()
    print(f"PFA_EXIT {os.environ['PFA_EXIT']} {1e3 * (time.time() - _t):.1f} ms", file=sys.stderr)
# The kernel's teardown of the process's GPU context (after the last instruction here) takes 2 ms or 60-100 ms: it waits for
# something that completes sooner when other CPUs pass through the scheduler at that moment — a process that dies with a few
# dozen sleeping threads (numpy's 256 idle OpenBLAS threads did it by accident) exits fast, a single-threaded one slowly.
# Measured from a shell, one file, three alternating rounds (profiles/r04_shell_cold_start_ab.txt): one OpenBLAS thread
# 217-359 ms (median 287), 256 threads 159-334 (median 230), one thread + 16-64 sleeping threads at exit 121-293 (median 165).
# PFA_EXIT_KICK=0 switches it off.
try:
    _kick = int(os.environ.get("PFA_EXIT_KICK", "32") or 0)
except ValueError:              # (a run that has succeeded does not fail over this)
    _kick = 0
if _kick > 0:
    try:
        import threading
        _ev = threading.Event()
        for _ in range(_kick):
            threading.Thread(target=_ev.wait, daemon=True).start()
    except Exception:           # noqa: BLE001  (a thread limit: the exit is only slower)
        pass
sys.stdout.flush()
sys.stderr.flush()
os._exit(rc or 0)
