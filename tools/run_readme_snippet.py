"""Run the Python block of README.md as it stands (keeps the README honest)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
code = re.search(r"```python\n(.*?)```", open(os.path.join(ROOT, "README.md")).read(), re.S).group(1)
exec(compile(code, "README.md", "exec"))
