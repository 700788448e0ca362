// k_ffn_rows for d_model 48: the kernel template of ffd_ffn_rows.hip instantiated in its own translation unit (compile time).
#define FFD_ROWS_EXTRA_D 48
#include "ffd_ffn_rows.hip"
