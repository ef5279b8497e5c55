#!/bin/bash
# Static instruction counts and register tables on the sources in the tree (no GPU): profiles/isa_counts.json (both bench
# expressions + the ahead-of-time kernels the bench's legs price) and profiles/r06_isa_counts_aot.json.  Run after ANY change to
# physicl_amd/csrc/* or the build options, before tools/final_run.sh (calibrations in isa_counts.json start over when the
# sources change: tools/summarize_driver_prof.py / summarize_calib_ahead.py write them back from the new profiles).
set -e
cd "$(dirname "$0")/.."
K=$(mktemp -d)
python tools/isa_count.py --json profiles/isa_counts.json > /dev/null
python tools/isa_count.py --json profiles/isa_counts.json "2.5E+25 * exp(r2[gid] / 8600.0)" > /dev/null
python tools/isa_count.py --json profiles/isa_counts.json --keep $K --aot "k_delete_ahead_live<double, false>;k_delete_ahead_live<double, true>;k_mixed<double, false, 0>;k_mixed<float, false, 0>;k_mixed3<double, false>;k_mixed3<float, false>;k_multi<double, false, 0>" > /dev/null
python tools/aot_spill_table.py --asm $K/aot.s
rm -rf $K
