# issue-rate micro-benchmark (each kernel is a few hundred microseconds; the whole run is bounded)
cd $GRAFT_REPO_ROOT/tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/issue_rates issue_rates.hip 2>/dev/null && timeout -k 5 60 /tmp/issue_rates
