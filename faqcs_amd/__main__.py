"""`python -m faqcs_amd [FaQCs flags]` -- the FaQCs process contract on the MI355X hot path."""
import sys

from .driver import run

if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
