# HIP side of the J&F noise floor on fixture G14: the default build with the stem weights moved by K = 0..7 ulp (the perturbation family of the
# oracle's own recorded draws), one dataset run each.   bash tools/jf_ensemble.sh [extra env assignments]
mkdir -p gpurun_out/jf_g14
for k in 0 1 2 3 4 5 6 7; do env "$@" JF_PERTURB=$k python tools/jf_g14.py ens_p$k${JF_TAG:-} 2>&1 | grep "J&F"; done
