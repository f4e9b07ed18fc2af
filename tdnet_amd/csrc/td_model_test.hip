// td_model_test.hip -- the translation unit of libtdnet_hip_test.so: everything libtdnet_hip.so exports (td_model.hip, the same sources) PLUS the
// single-operator entry points and tuning probes of include/tdnet_test.h (td_ops_test.h).  Loaded by tests/ and tools/ only (tdnet_amd/_capi.py
// test_lib()); the product path (tdnet_amd/model/*, engine.py, bench.py's timed loops) loads libtdnet_hip.so, which has no tdnet_op_* / tdnet_bench_*.
#include "td_model.hip"
#include "td_ops_test.h"
