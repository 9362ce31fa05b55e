import sys

from .config import main

sys.exit(main())
