#!/bin/bash
# A/B of two prebuilt libraries on one box: stage times of the headline view (the tests run on the new one)
cp profiles/tools/ab/lib_new.so vegs_amd/_lib/libvegsrast.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_c_harness.py tests/test_gpu_views.py tests/test_gpu_render_all.py -q -x 2>&1 | tail -3
VEGS_FUZZ_SEEDS=0:300 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2
for v in old new old new; do
  cp profiles/tools/ab/lib_$v.so vegs_amd/_lib/libvegsrast.so
  echo "== $v"; python bench.py --stages --no-variants --no-cpu-baseline 2>&1 | grep "stage breakdown\|ms_per_step" | sed 's/.*breakdown//' | cut -c1-420
done
cp profiles/tools/ab/lib_new.so vegs_amd/_lib/libvegsrast.so
