# Dev helper (GPU box): build with clock stamps, print the cull kernel's phase times on configs[1] / configs[2], rebuild clean
touch livescan3d_amd/csrc/icp.hip
make -C livescan3d_amd/csrc -j12 EXTRA="-DLSN_CULL_STAMPS $AB_EXTRA" > /dev/null 2>&1 || { echo build failed; exit 1; }
for sens in 2 8; do for near in 2 1 0; do echo "== sensors=$sens near=$near"; ICP_SENSORS=$sens LSN_ICP_NEAR=$near timeout -k 10 120 python3 tools/cull_stamps.py 2>&1 | tail -9; done; done
touch livescan3d_amd/csrc/icp.hip; make -C livescan3d_amd/csrc -j12 > /dev/null 2>&1
