// The lean instances of the fused half-steps at the full geometry with STREAMED lists (non-temporal loads: mu_ell_kernel.hpp,
// ell_list_load; espm_mu_state.ell_stream): mu_fused_plain.hip once more under ESPM_PLAIN_STREAM, as launch_fused_plain_stream.
#define ESPM_PLAIN_STREAM 1
#include "mu_fused_plain.hip"
