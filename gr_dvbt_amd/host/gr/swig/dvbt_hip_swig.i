/* dvbt_hip_swig.i -- the lines gr-dvbt's swig/dvbt_swig.i gains for the two blocks of this directory that gr-dvbt does not declare itself
 * (pattern: swig/dvbt_swig.i:14-31 includes, :33-51 %include, :53-71 GR_SWIG_BLOCK_MAGIC2).  The ten shells that keep a reference class name
 * (ofdm_sym_acquisition ... energy_descramble) need nothing: their headers, SWIG lines and grc/*.xml are gr-dvbt's own.
 *
 *   %{                                   // into the %{ ... %} block of dvbt_swig.i
 *   #include "dvbt/fft_hip.h"
 *   #include "dvbt/rx_hip.h"
 *   %}
 */
%include "dvbt/fft_hip.h"
%include "dvbt/rx_hip.h"
GR_SWIG_BLOCK_MAGIC2(dvbt, fft_hip);
GR_SWIG_BLOCK_MAGIC2(dvbt, rx_hip);
