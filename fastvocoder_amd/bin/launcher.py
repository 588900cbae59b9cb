"""``MODE=<synthesize|test|publish> python -m fastvocoder_amd.bin.launcher --flags``
-- the reference's $MODE dispatch (bin/launcher.py:7-19) for the inference-side
modes.  ``train`` / ``preprocess`` are training-side and out of scope."""
import os
import sys


def main():
    mode = os.getenv("MODE")
    if mode == "synthesize":
        from .synthesize import run_synthesizer
        run_synthesizer()
    elif mode == "test":
        from .test import run_test
        run_test()
    elif mode == "publish":
        from .publish import run_publisher
        run_publisher()
    elif mode in ("train", "preprocess"):
        sys.exit(f"MODE={mode} is a training-side mode of the reference and is not part of "
                 "fastvocoder_amd (generator inference only)")
    else:
        sys.exit("set MODE=synthesize | test | publish")


if __name__ == "__main__":
    main()
