cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "many_strains or selected_genome or file_bitmaps or planes_are_clean or k31 or multi_sequence or reverse_complement" 2>&1 | tail -25
