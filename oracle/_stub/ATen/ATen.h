// intentionally empty: stands in for <ATen/ATen.h> when compiling the reference's dcn_v2_im2col_cpu.cpp (it uses no ATen symbol)
