mkdir -p gpurun_out/r6full
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r6full/pytest.log 2>&1; tail -8 gpurun_out/r6full/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
