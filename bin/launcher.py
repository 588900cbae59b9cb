"""Repo-root shim so the reference's command line works unchanged:
``MODE=synthesize python3 bin/launcher.py --checkpoint_path ... --mel_path ...``"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from fastvocoder_amd.bin.launcher import main
    main()
