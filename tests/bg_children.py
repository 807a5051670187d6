"""Child processes of the GPU suite, run in the background.

Some tests need a process of their own: a switch the library reads once per process (NPP_H3_CFG, NPP_EPI_LEAN, NPP_G8_MAXK ...), a
pluggable allocator (tests/test_poison_gpu.py), a process group, an unchanged launcher script.  Run one after the other each of them
pays its own interpreter start, `import torch` and GPU start-up while the pytest process sits idle: 28 children, ~250 s of a 470 s
suite.  They are independent of the pytest process, so: a test module REGISTERS its children when it is imported; the first test
that asks for a result starts ALL registered children (two at a time, in registration order = the order the tests run in) and every
test then waits only for its own.  The pytest process goes on with its in-process GPU tests meanwhile.  At most 2 children + the
pytest process + the 2 ranks a multi-process test spawns touch the GPU together (the box allows 6)."""
import concurrent.futures
import os
import subprocess

_specs = {}        # key -> (argv, env, timeout)
_futures = {}
_pool = None
_before = []       # callables run once, before the first child starts (e.g. build a helper library every probe loads)


def before_start(fn):
    if fn not in _before:
        _before.append(fn)


def register(key, argv, env=None, timeout=900):
    """`env`: extra variables.  Every child gets a rendezvous port of its own (two children with a process group may run together)."""
    if key in _specs:
        return key
    full = dict(os.environ)
    full["MASTER_PORT"] = str(29700 + len(_specs))
    full.update(env or {})
    _specs[key] = (list(argv), full, timeout)
    return key


def _start_all():
    global _pool
    while _before:
        _before.pop(0)()
    _pool = concurrent.futures.ThreadPoolExecutor(max_workers=int(os.environ.get("NPP_TEST_CHILDREN", "2")))
    for key, (argv, env, timeout) in _specs.items():
        if key not in _futures:
            _futures[key] = _pool.submit(subprocess.run, argv, env=env, capture_output=True, text=True, timeout=timeout)


def result(key):
    """The finished child `key` (subprocess.CompletedProcess); starts every registered child on first use."""
    if key not in _futures:
        _start_all()
    return _futures[key].result()
