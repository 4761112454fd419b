// host/reader_dev.h — the device-side FASTQ splitter (hip/fastq_dev.hip) as host/reader.cpp sees it
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../../include/salmon_hip.h"
struct sq_dev_reader;
int sq_dev_reader_open(const std::vector<std::string>& files1, const std::vector<std::string>& files2, uint32_t batch, uint32_t slots, sq_dev_reader** out);   // SQ_ERR_DEVICE: no device, keep the host path
int sq_dev_reader_next(sq_dev_reader*, sq_read_batch* b, int* slot);
void sq_dev_reader_release(sq_dev_reader*, int slot);
uint64_t sq_dev_reader_total(const sq_dev_reader*);
void sq_dev_reader_close(sq_dev_reader*);
