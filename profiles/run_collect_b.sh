#!/bin/bash
PART=b bash profiles/collect_round.sh r04
