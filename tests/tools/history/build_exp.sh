# measurement builds of the library with experiment switches (results are wrong; timing only): libvmapstep_<name>.so
set -e
cd "$(dirname "$0")/../.."
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -D$V -I vmap_amd/csrc vmap_amd/csrc/vmapstep.hip -o vmap_amd/libvmapstep_$V.so &
done
wait
