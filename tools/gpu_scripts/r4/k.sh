#!/bin/bash
# round 4: the new tests (mixture functor, full-size properties of configs 3 / 5, two half-batches with a user functor, bench
# line extras, shared-GPU refusal), then kernel stats of configs 5 and 3
O=$PWD/gpurun_out/r4k; mkdir -p $O
timeout -s KILL 900 python -m pytest -m gpu -q -x \
  "tests/test_gpu_sample_correctness.py::test_reference_mixture_of_two_normals_through_a_device_functor" \
  "tests/test_gpu_configs.py::test_config3_full_size_properties" "tests/test_gpu_configs.py::test_config5_full_size_properties" \
  "tests/test_gpu_user_functor.py::test_user_functor_with_the_dense_metric_is_bit_equal_to_the_builtin_family" \
  "tests/test_gpu_statistics.py::test_two_step_objects_over_one_context_do_not_trust_a_stale_position" \
  "tests/test_gpu_configs.py::test_bench_launches_its_own_ranks" \
  "tests/test_gpu_configs.py::test_bench_default_line_carries_the_other_configs_and_a_live_traffic_figure" 2>&1 | tail -25 | tee $O/pytest.log
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
for c in 5 3; do
  rm -rf /tmp/pk$c; rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/pk$c -o t -- python $REPO/bench.py --config $c --steps 3 --warmup 1 > $O/bench_c${c}_prof.json 2> /tmp/pk$c.err
  f=$(find /tmp/pk$c -name '*kernel_stats.csv' | head -1); cp $f $O/c${c}_kernel_stats.csv
  python - <<PY
import csv
rows = list(csv.DictReader(open('$f')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('config $c: kernel time %.1f ms' % (tot / 1e6))
for r in rows[:12]:
    print('  %5.1f%%  calls %6s  avg %9.1f us  %s' % (float(r['Percentage']), r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:90]))
PY
done
