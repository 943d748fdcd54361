/* Plain-C consumer of include/pfa.h: proves the boundary is a C ABI (no C++ or
 * torch types) and that the host entry points work without a GPU. */
#include <stdio.h>
#include <string.h>

#include "../include/pfa.h"

int main(void) {
  if (pfa_abi_version() != PFA_ABI_VERSION) return 1;
  /* ATG vs ATA: codes A=0,C=1,T=2,G=3 -> ATG = 0*16+2*4+3 = 11, ATA = 8; third position differs, M vs I: non-synonymous */
  if (pfa_classify_codon_mask((1ull << 11) | (1ull << 8)) != (4u << 3)) return 2;
  if (pfa_codon_syn_thirds(11) != 0) return 3;
  const char* fa = ">a\nacgt\n>b\nACGA\n";
  pfa_fasta_rec recs[4];
  if (pfa_fasta_index((const uint8_t*)fa, strlen(fa), recs, 4) != 2) return 4;
  uint8_t mat[32];
  if (pfa_fasta_pack((const uint8_t*)fa, strlen(fa), recs, 2, mat, 16) != PFA_OK) return 5;
  if (memcmp(mat, "ACGT", 4) || memcmp(mat + 16, "ACGA", 4)) return 6;
  if (pfa_device_count() == 0) {
    if (pfa_init(0) != PFA_ERR_NO_DEVICE) return 7;
    if (!strstr(pfa_last_error(), "no CPU fallback")) return 8;
  }
  printf("abi ok\n");
  return 0;
}
