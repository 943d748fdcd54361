import os
import sys
import time

_t_enter = time.time()          # wall clock: bench.py's cold_start leg subtracts its own spawn time

# A one-file run is dominated by start-up (interpreter, numpy, hipInit: bench.py `cold_start`).  The
# HIP runtime is brought up by a native thread WHILE Python imports numpy and parses the command line;
# pfa_init() joins it.  Ranks of a multi-GPU job bind their CPUs first and initialise in pfa_init().
if "WORLD_SIZE" not in os.environ and not os.environ.get("PFA_NO_PREWARM"):
    try:
        import ctypes
        _lib_path = os.environ.get("PFA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpfa_hip.so")
        ctypes.CDLL(_lib_path).pfa_prewarm(int(os.environ.get("LOCAL_RANK", "0")))
    except (OSError, AttributeError, ValueError):
        pass        # cli.main() reports a missing library in its own words

from . import cli  # noqa: E402

cli.startup_marks.append(("interpreter+package+prewarm start", _t_enter))
cli.startup_marks.append(("import cli (numpy, ctypes, argparse)", time.time()))
try:
    rc = cli.main()
except SystemExit as e:         # argparse errors, --help
    rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    if e.code is not None and not isinstance(e.code, int):
        print(e.code, file=sys.stderr)
# every row has been written: leave without the interpreter's and the HIP runtime's teardown (45 ms of a
# 0.3 s one-file run); the driver releases the device when the process is gone
sys.stdout.flush()
sys.stderr.flush()
os._exit(rc or 0)
