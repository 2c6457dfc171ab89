# Per-kernel-form ablation of the dataset-level J&F on fixture G14 (VERDICT r3 "Next" #1).   bash tools/jf_ablate.sh
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python tools/jf_g14.py $tag 2>&1 | grep -v amdgpu.ids | tail -n 1; }
run default                     X=1
run no_wino6                    FRTM_NO_WINO6=1
run no_wino4_wino6              FRTM_NO_WINO4=1
run no_winograd_at_all          FRTM_NO_WINO4=1 JF_NO_WINOGRAD=1
run no_resident_joint_fit       FRTM_NO_PERSISTENT_JOINT=1
run no_persistent_cg            JF_NO_PERSISTENT_CG=1
run no_windows                  JF_NO_WINDOWS=1
run direct_chain_no_windows     FRTM_NO_WINO4=1 JF_NO_WINOGRAD=1 FRTM_NO_PERSISTENT_JOINT=1 JF_NO_PERSISTENT_CG=1 JF_NO_WINDOWS=1
