"""Which build of the C-ABI library ``_lib`` binds.

The product binds ``lib/libtcvn_hip.so`` and nothing in the environment can change that.  The variant-vs-variant tests need the
``-DTCVN_DEBUG_KNOBS`` build (``libtcvn_hip_dbg.so``: same sources, validation switches honoured) in a child process: the child calls
``use("libtcvn_hip_dbg.so")`` explicitly BEFORE ``transformercvn.hip._lib`` is imported.  bench.py refuses to run on anything but the
product library and records the file it loaded."""
NAME = "libtcvn_hip.so"
_bound = False


def use(name: str) -> None:
    global NAME
    if _bound:
        raise RuntimeError("transformercvn.hip._lib is already bound to " + NAME)
    # product library, the validation build, or a kept copy of an earlier validation build (libtcvn_hip_dbg_<tag>.so: same-box A/B timing,
    # tools/time_dbg.py TIME_LIB=...)
    if name != "libtcvn_hip.so" and not (name.startswith("libtcvn_hip_dbg") and name.endswith(".so") and "/" not in name):
        raise ValueError(name)
    NAME = name
