"""Calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on known byte counts in OUR access patterns
(guide: MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of a wide coalesced stream; other
widths are uncalibrated).  Runs three torch kernels with known traffic:
  copy16   : 16 B/lane coalesced read+write of 1 GiB  (torch copy_)
  read4    : 4 B/lane coalesced read of 1 GiB         (int32 sum)
  gather4  : 64 M random 4-byte gathers from a 1 GiB table (index_select)
The profile script divides FETCH_SIZE by the known bytes of each."""
import torch
n = 1 << 28  # 2^28 int32 = 1 GiB
a = torch.arange(n, dtype=torch.int32, device="cuda")
b = torch.empty_like(a)
idx = torch.randint(0, n, (1 << 26,), device="cuda", dtype=torch.int64)
torch.cuda.synchronize()
for _ in range(2):
    b.copy_(a)
    s = a.sum()
    g = a.index_select(0, idx)
torch.cuda.synchronize()
print("known bytes: copy 2^30 read + 2^30 write; sum 2^30 read; gather 2^26 * 4 B useful (64 B sectors: 2^32)")
