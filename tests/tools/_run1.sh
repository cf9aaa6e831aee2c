R01=drl-on-robot-arm_amd/build/r01/tests/tools/time_rollout.py
NEW=tests/tools/time_rollout.py
V=$PWD/drl-on-robot-arm_amd/build/var
for rep in 1 2; do
python $R01 --T 100 --pre 600 --launches 20
ARMENV_LIB=$V/libarmenv_e1.so python $NEW --T 100 --pre 600 --launches 20
python $NEW --T 100 --pre 600 --launches 20
python $NEW --T 100 --pre 600 --launches 20 --set fence_counters=1
python $R01 --T 20 --pre 600 --launches 100
ARMENV_LIB=$V/libarmenv_e1.so python $NEW --T 20 --pre 600 --launches 100
python $NEW --T 20 --pre 600 --launches 100
python $R01 --step-api --pre 600 --launches 5
ARMENV_LIB=$V/libarmenv_e1.so python $NEW --step-api --pre 600 --launches 5
python $NEW --step-api --pre 600 --launches 5
done
