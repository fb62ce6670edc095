# A/B of the solve kernel's record reduction, ON THE GPU BOX (rebuilds the library): bash scripts/experiments/solve_nb_ab.sh
cd $GRAFT_REPO_ROOT
for cfg in "256 32" "512 32" "512 16" "128 32"; do
  set -- $cfg
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fPIC -shared -DTC_SOLVE_NT=$1 -DTC_SOLVE_NB=$2 tightly_coupled_sfm_amd/csrc/tcsfm_api.hip -o tightly_coupled_sfm_amd/libtcsfm_hip.so
  echo "threads=$1 loads_per_batch=$2"; python scripts/diag/solve_stamps.py 2>&1 | tail -2
done
