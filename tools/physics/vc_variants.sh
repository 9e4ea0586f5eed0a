# Diagnostic variants of the ANYmal value calibration (tools/physics/value_calibration.py); the asserted run is the first line.
for v in "" "VC_DET=1" "VC_CMD=0.3,1.0,0,0,0,0" "VC_CMD=0.3,1.0,0,0,0,0 VC_DET=1" "VC_CMD=-1,1,0,0,0,0" "VC_CMD=0.3,1.0,-0.3,0.3,-0.5,0.5" "VC_CMD=0,0,0,0,0,0"; do
  echo "== $v"
  env $v python tools/physics/value_calibration.py anymal 4096 700 2>/dev/null | sed -n "/^{/,\$p" | python -c "
import json,sys; d=json.load(sys.stdin)['anymal_c_flat']; s=d['steady']
print('steady V %.3f G %.3f bias %.3f r %.3f | startup fall %.3f steady falls/step %.5f rew/step %.5f' % (s['mean_V'], s['mean_G'], s['bias'], s['pearson_r'], d['startup_fall_rate'], d['steady_falls_per_env_step'], d['mean_reward_per_step']))"
done
