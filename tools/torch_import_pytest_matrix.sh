#!/bin/bash
out=gpurun_out/$1; mkdir -p $out; LIMIT=${2:-60}
export PYTHONPATH=$PWD
mkdir -p /tmp/pt && cd /tmp/pt
cat > test_a.py <<'PY'
import time
def test_devcount_then_torch():
    from helib_amd import capi
    print(capi.device_count())
    t0 = time.time(); import torch; print("torch imported in", round(time.time() - t0, 1))
PY
cat > test_b.py <<'PY'
import time
def test_context_then_torch():
    from helib_amd import capi as hx
    c = hx.Context(16384, 0)
    t0 = time.time(); import torch; print("torch imported in", round(time.time() - t0, 1))
PY
cat > test_c.py <<'PY'
import time
def test_torch_only():
    t0 = time.time(); import torch; print("torch imported in", round(time.time() - t0, 1))
PY
run() { local t0=$SECONDS; timeout $LIMIT "$@" > $OLDPWD/$out/$NAME.log 2>&1; echo "$NAME rc=$? $((SECONDS-t0)) s: $(grep -h "torch imported" $OLDPWD/$out/$NAME.log | head -1)"; }
ORDER=${3:-1}
if [ "$ORDER" = 2 ]; then
NAME=a_devcount_noplugins run python -m pytest -q -s -p no:hypothesis -p no:timeout -p no:xdist -p no:anyio -p no:cacheprovider test_a.py
NAME=a_devcount run python -m pytest -q -s test_a.py
NAME=a_devcount_again run python -m pytest -q -s test_a.py
ls -la ~/.cache 2>/dev/null | head; du -sh ~/.cache/* 2>/dev/null | head
else
NAME=c_torch_only run python -m pytest -q -s test_c.py
NAME=a_devcount run python -m pytest -q -s test_a.py
fi
