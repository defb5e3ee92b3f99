// Shared by the translation units of libmgn_hip.so.
#pragma once
#include "../../include/mgn_hip.h"

void mgn_set_error(const char* fmt, ...);

// 0: fp32 MFMA everywhere; 1: three-term bf16 split in the edge kernel; 2 (default): two-term fp16 split.
// Chosen once per process by the environment variable MGN_MATMUL = f32 | bf16x3 | f16x2.
int mgn_matmul_mode();
void mgn_set_matmul_mode_internal(int mode);

// range flag of the training translation unit (see mgn_range_check)
int mgn_train_read_range_flag(int* tripped, int reset, void* stream);
int mgn_train_poll_range_flag(int* host_word, void* stream);
