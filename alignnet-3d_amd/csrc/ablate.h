// Ablation and timing hooks of the kernels: result-changing switches (skip the Gram MFMAs, a store, a whole lift ...) and cycle stamps used
// to find out where a kernel's time goes.  They exist ONLY in the ablation build (`make ablate` -> libalignnet_hip_ablate.so,
// -DALIGNNET_ABLATE; tools/ab_build.sh, tools/time_train_kernels.py); in the shipped library every test below is the constant 0 and the
// code behind it is not compiled, and the library reads no environment variable at all (tests/test_capi_cpu.py checks both).
#pragma once
#ifdef ALIGNNET_ABLATE
#define ALN_ABL(flags, bits) ((flags) & (bits))
#define ALN_STAMPS(p) (p)
#else
#define ALN_ABL(flags, bits) 0
#define ALN_STAMPS(p) (static_cast<long long*>(nullptr))
#endif
