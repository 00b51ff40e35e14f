set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02u; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/resident_probe.py --rows 1000000 --cols 8 --estimators 300 --categorical 2>&1 | grep "^run" | tee $O/resident_probe_categorical.log
timeout 300 python tools/resident_probe.py --rows 1000000 --cols 8 --estimators 300 2>&1 | grep "^run" | tee $O/resident_probe_object.log
timeout 300 python -m pytest tests/test_resident_path.py -x -q -m gpu 2>&1 | tail -3
