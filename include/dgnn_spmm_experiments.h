/*
 * dgnn_spmm_experiments.h -- A/B knobs of the benchmark tools. NOT part of the product ABI: the symbols exist only in a
 * library built with -DDGNN_EXPERIMENTS (dist-gnn_amd/lib/libdgnn_spmm_exp.so, built by __graft_entry__.build() beside
 * the product library; tools/*_bench.py and one test load it through DGNN_LIB). No counterpart in the reference.
 */
#ifndef DGNN_SPMM_EXPERIMENTS_H
#define DGNN_SPMM_EXPERIMENTS_H
#ifdef __cplusplus
extern "C" {
#endif
/* 0 = the product's dispatch. Other values pick lane-group widths, gathers in flight, index widths, or switch parts of a
 * kernel off for timing (see the dispatchers in csrc/dgnn_spmm.hip). Process-global, not thread-safe. */
int dgnn_set_variant(int variant);
int dgnn_get_variant(void);
#ifdef __cplusplus
}
#endif
#endif /* DGNN_SPMM_EXPERIMENTS_H */
