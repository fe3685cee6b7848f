import sys

from .profile import main

sys.exit(main())
