// Private layout of the exchange context (mu_xchg.hip, mu_api.hip); the public type is opaque (include/espm_mu.h).
#pragma once
#include <stddef.h>

struct espm_xchg {
  int world, rank;
  size_t record_bytes, mailbox_bytes, off_flags, off_err;
  size_t off_wgflags;            // [world][wgflags] uint32: flags of the fused exchange (one per reduction workgroup of a rank + one)
  int wgflags;
  size_t off_gran;               // [2][world][34 wgflags + 2 HS_STRIDE] x 8 bytes: granules {value bits, sequence number} of the in-launch exchange
                                 // (parity = seq & 1): 32 per reduction workgroup for its piece of A, then (from 32 wgflags on) 2 per workgroup
                                 // for its row sum, then (from 34 wgflags on) 2 per statistic of the rank's new H block
  unsigned char* mailbox;        // this rank's mailbox (device)
  unsigned char* staging;        // where the rank packs its record before post (device)
  unsigned char* peers[16];      // mailbox of every rank as mapped here (peers[rank] == mailbox)
  bool opened[16];
  int order;                     // 0: flags as relaxed stores behind the data's drain (default); 1: release stores at system scope (espm_xchg_set_order, ESPM_XCHG_ORDER=release)
};
