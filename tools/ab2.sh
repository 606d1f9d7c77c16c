#!/bin/bash
# same-box A/B of two builds of the library (ab_libs/libA.so, ab_libs/libB.so): alternating runs of the timed region; prints the step rate and the
# stand-alone times of the kernels named as arguments
KS=${@:-orb_fast_kernel orb_pyramid_kernel}
run() { cp ab_libs/lib$1.so corb-slam_amd/libcorb_accel.so; timeout 300 python bench.py --cpu-frames 0 --ba-cpu-kf 0 --ba-kf 0 --replay-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', d['value'], ' '.join('%s alone %.1f' % (k, r['alone_unsplit_avg_us'][k]) for k in '$KS'.split()))"; }
cp corb-slam_amd/libcorb_accel.so /tmp/lib_keep.so
for i in 1 2 3; do run A; run B; done
cp /tmp/lib_keep.so corb-slam_amd/libcorb_accel.so
