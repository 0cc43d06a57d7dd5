#!/bin/bash
# full GPU suite + smoke, as the driver runs them at round end
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-tests}; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
