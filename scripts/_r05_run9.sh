set -u
mkdir -p gpurun_out/r05i
timeout 1500 python -m pytest tests/test_reference_events_gpu.py tests/test_reference_replay_gpu.py tests/test_small_gpu.py tests/test_copies_gpu.py tests/test_select_gpu.py tests/test_hybrid_gpu.py -q -k "not test_small_form_matches" > gpurun_out/r05i/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05i/tests.log
tail -30 gpurun_out/r05i/tests.log
