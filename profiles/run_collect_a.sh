#!/bin/bash
PART=a bash profiles/collect_round.sh r04
