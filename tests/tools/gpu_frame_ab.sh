mkdir -p gpurun_out/abl; : > gpurun_out/abl/frame_ab.jsonl
for rep in 1 2 3; do for lib in tests/tools/libvmapstep_base.so vmap_amd/libvmapstep.so tests/tools/libvmapstep_n96.so; do
VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python tests/tools/frame_bench.py 2>/dev/null | grep "^{" | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'lib':'$lib','obj':j['objects_ms_per_frame'],'bg':j['background_ms_per_frame'],'both_serial':j['objects_plus_background_ms_per_frame'],'two_streams':j['objects_and_background_on_two_streams_ms_per_frame']}))" >> gpurun_out/abl/frame_ab.jsonl
done; done; cat gpurun_out/abl/frame_ab.jsonl
