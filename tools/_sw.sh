python tools/tol_probe.py survey 2>&1 | grep tolerance
DMG_DEEP_ROWS=160 python tools/tol_probe.py survey 2>&1 | grep tolerance
python tools/tol_probe.py ocean 2>&1 | grep tolerance
