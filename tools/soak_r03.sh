#!/bin/bash
# round 3 soak on the final build: the seeded fuzzers at many times their default sizes (device geometries, the Python surface, pairwise cohorts,
# adversarial VCF text through the whole binary), once on packed matrices and the device fuzzers once more on u8 rows with a tiny grid and deep deferral
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/soak_r03
mkdir -p $O
FERROMIC_FUZZ_DEVICE_CASES=1500 timeout -k 10 500 python -m pytest tests/test_gpu_device_fuzz.py -x -q > $O/device_fuzz.log 2>&1; echo "device fuzz exit $?"; tail -1 $O/device_fuzz.log
FERROMIC_FUZZ_CASES=4000 FERROMIC_FUZZ_NUMPY_CASES=2000 FERROMIC_FUZZ_PAIR_CASES=300 timeout -k 10 800 python -m pytest tests/test_gpu_api_fuzz.py -x -q > $O/api_fuzz.log 2>&1; echo "api fuzz exit $?"; tail -1 $O/api_fuzz.log
FERROMIC_FUZZ_PIPELINE_CASES=60 timeout -k 10 600 python -m pytest tests/test_gpu_run_vcf.py -x -q -k adversarial > $O/pipeline_fuzz.log 2>&1; echo "pipeline fuzz exit $?"; tail -1 $O/pipeline_fuzz.log
FMH_LAYOUT=bytes FMH_GRID_BLOCKS=1 FMH_DEFER_TILES=16 FERROMIC_FUZZ_DEVICE_CASES=300 timeout -k 10 500 python -m pytest tests/test_gpu_device_fuzz.py tests/test_gpu_api_fuzz.py -x -q > $O/u8_one_block.log 2>&1; echo "u8 one-block exit $?"; tail -1 $O/u8_one_block.log
FMH_GRID_BLOCKS=2 FMH_PIPE=1 timeout -k 10 600 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_api_fuzz.py tests/test_gpu_api_dropin.py tests/test_gpu_comm.py -x -q > $O/two_blocks_pipe_everywhere.log 2>&1; echo "two blocks, pipelined loop wherever built: exit $?"; tail -1 $O/two_blocks_pipe_everywhere.log
FMH_PACKED_NO_PREFETCH=1 timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_device_fuzz.py tests/test_gpu_api_fuzz.py tests/test_gpu_comm.py -x -q > $O/no_prefetch_row_loops.log 2>&1; echo "row loops of several trips everywhere (eight groups: shallow batches): exit $?"; tail -1 $O/no_prefetch_row_loops.log
FMH_PIPE=0 FMH_DEFER_TILES=1 timeout -k 10 600 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_device_fuzz.py -x -q > $O/plain_tile_loop.log 2>&1; echo "plain tile loop, nothing deferred: exit $?"; tail -1 $O/plain_tile_loop.log
FERROMIC_TRACK_WRITER=runs timeout -k 10 600 python -m pytest tests/test_gpu_run_vcf.py -x -q > $O/run_vcf_run_aware_writer.log 2>&1; echo "run_vcf with the run-aware writer on every track: exit $?"; tail -1 $O/run_vcf_run_aware_writer.log
FERROMIC_NUMPY_BYTES=1 timeout -k 10 600 python -m pytest tests/test_gpu_api_dropin.py tests/test_gpu_api_fuzz.py -x -q > $O/from_numpy_byte_route.log 2>&1; echo "from_numpy on the byte route: exit $?"; tail -1 $O/from_numpy_byte_route.log
