mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_binding_replay.py tests/test_torch_ops.py tests/test_autograd.py -m gpu -q -x -v > gpurun_out/r3_exp9_pytest.log 2>&1
grep -n "PASSED\|FAILED\|Fatal\|fault\|Error" gpurun_out/r3_exp9_pytest.log | head -30
grep -n "File \"/root/repo\|File \"/tmp/code" gpurun_out/r3_exp9_pytest.log | head -20
