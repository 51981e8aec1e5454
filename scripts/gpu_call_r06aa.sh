#!/bin/bash
# round 6, last session: features of the linear part alone on the HIP training step — the fit-fuzz seeds (of 0 .. 899) that hold such
# features, the in-suite fit seeds, and the training / fit / rank-path tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aa; mkdir -p $O
export DCTR_FUZZ_SEEDS=1 DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_FIT_SEEDS=$(python - <<'PY'
lo=[14, 28, 42, 49, 91, 112, 119, 161, 182, 189, 203, 217, 238, 252, 266, 273, 294, 301, 306, 309, 312, 314, 318, 322, 328, 329, 331, 332, 337, 342, 343, 344, 349, 351, 352, 359, 369, 379, 388, 389, 391, 393, 394, 396, 399, 402, 407, 408, 412, 417, 419, 422, 423, 427, 434, 437, 438, 446, 449, 452, 459, 461, 462, 467, 469, 473, 477, 479, 482, 486, 489, 491, 494, 499, 503, 509, 512, 517, 518, 521, 522, 526, 532, 534, 538, 543, 547, 552, 553, 556, 557, 558, 561, 562, 566, 568, 573, 574, 579, 592, 593, 594, 599, 601, 604, 606, 609, 617, 619, 621, 622, 623, 624, 626, 628, 629, 638, 641, 642, 643, 644, 646, 647, 648, 649, 651, 654, 658, 659, 667, 672, 674, 679, 682, 687, 689, 693, 697, 699, 703, 717, 721, 723, 724, 726, 727, 729, 733, 734, 738, 743, 744, 749, 758, 762, 763, 766, 769, 772, 773, 774, 779, 787, 794, 796, 799, 806, 812, 816, 817, 818, 827, 828, 831, 832, 833, 834, 838, 841, 842, 844, 847, 848, 852, 853, 854, 862, 868, 872, 873, 874, 878, 879, 883, 886, 896, 899]
print(','.join(str(i) for i in sorted(set(lo) | set(range(160)))))
PY
)
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rfs -k "trains_alike" > $O/pytest_fitfuzz.log 2>&1
tail -1 $O/pytest_fitfuzz.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_fitfuzz.log | cut -c1-300 | head -30
grep "^SKIPPED" $O/pytest_fitfuzz.log | sed 's/fit fuzz [0-9]* //; s/(.*//' | cut -c1-120 | sort | uniq -c | sort -rn | head -8
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_train.py tests/test_gpu_train_reg.py tests/test_gpu_din_train.py tests/test_gpu_rank_path.py tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest_train.log 2>&1
tail -1 $O/pytest_train.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_train.log | cut -c1-300 | head -20
