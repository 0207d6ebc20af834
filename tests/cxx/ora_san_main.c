/* ora_san_main.c -- runs the oracle receiver (the parity checker itself) under ASan + UBSan on a cf32 IQ file:
 * usage: ora_san <iq.cf32> <n_subch> ; sub-channels are the canonical i*48 CU / 64 kbit/s / EEP 3-A layout. */
#include <stdio.h>
#include <stdlib.h>
#include "../../oracle/dab_oracle.h"

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  ora_cf32 *x = (ora_cf32 *)malloc((size_t)bytes);
  if (fread(x, 1, (size_t)bytes, f) != (size_t)bytes) return 2;
  fclose(f);
  const int n = atoi(argv[2]);
  ora_subch_desc d[64];
  for (int i = 0; i < n; i++) { d[i].subch_id = i; d[i].cu_start = 48 * i; d[i].cu_size = 48; d[i].kbps = 64; d[i].prot_level = 2; d[i].short_form = 0; }
  ora_receiver *rx = ora_rx_create(d, n);
  const int frames = ora_rx_run(rx, x, (size_t)bytes / sizeof(ora_cf32), 100000);
  const ora_rx_capture *cap = ora_rx_get_capture(rx);
  int ok = 0;
  for (int i = 0; i < frames * 12; i++) ok += cap->fib_crc[i];
  ora_cf32 tii[2048];
  const int ntii = ora_rx_take_tii(rx, tii);
  printf("ora_san ok: %d frames, %d FIBs with good CRC, %d TII null symbols\n", frames, ok, ntii);
  ora_rx_destroy(rx);
  free(x);
  return 0;
}
